"""ISA invariants of the hand-scheduled kernels, asserted on the disassembly of every build (build.py calls check_unit after hipcc).

The chains (mlphf.hip, mlpbf.hip) and the slotted k_dw_bf<6> stage (dw.hip) rely on instruction ORDER that hipcc does not know about: fragment
reads issued by inline asm straight into AGPRs (invisible to the compiler's s_waitcnt insertion), LDS-DMA pieces whose arrival is
published by a COUNTED `s_waitcnt vmcnt(16)` that assumes exactly the 16 tile stores of a k-step are younger than the last piece,
MFMAs tied to issue slots.  A later edit or a compiler upgrade that re-orders them corrupts results silently or loses the schedule;
round 3 met both.  Here the build fails instead:

  mlpbf.o, mlphf.o
           (a) every `ds_read_b128 a[..]` is retired by an `s_waitcnt` (lgkmcnt(0), or a counted lgkmcnt(N) that leaves it among the retired ones) before
               the first v_mfma that reads that AGPR.  Round 6: this rule refused a build in which hipcc had moved a slot's MFMA in front of the bare
               wait at the head of its slot group (an MFMA has no dependency on a wait) — the waits now carry the fragment register as an operand;
           (b) between the last `global_load_lds` and each counted publish `s_waitcnt vmcnt(K)` (K = 16 in mlpbf.o, 8 in mlphf.o) lie exactly K
               buffer_store_dword and no other vector-memory instruction;
  dw.o     (c) the main loop of k_dw_bf<6> holds 192 MFMAs (two slotted 8x8 stages), never more than two back to back, two counted
               `s_waitcnt vmcnt(8)` + `s_barrier`, and 16 LDS-DMA pieces each directly behind its m0 write and one wait state;
  every unit: no scratch_ instruction (build.py's remark check, restated on the disassembly);
           (d) no v_mfma reads, as SrcA or SrcB, a VGPR that a VALU instruction wrote fewer than TWO wait states earlier.  gfx950 needs
               them (measured, tools/hazardprobe.hip: with 0 or 1 wait states between `v_max_f32 vX, ...` and `v_mfma ... vX` the MFMA reads
               the OLD content of vX in > 96 % of the lanes, with 2 never); hipcc inserts them for the VALU instructions it can see and
               NOT behind one inside an asm statement.  The chains' element-wise ops (af_relu = one asm v_max_f32, bf_mask_keep) are such
               writers: sunk by the scheduler to just in front of the output layer's v_mfma_f32_4x4x1 that read them, they corrupted the
               forward of two-layer nets in round 3 (five such pairs in today's build with -DAF_NO_ELEMWISE_FENCE, none in the shipped one).
               AF_ELEMWISE_FENCE() (a sched_barrier behind the asm ops) keeps them apart; THIS rule is what proves it did, on every build.
"""
import os
import re
import shlex
import shutil
import subprocess
import tempfile

TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def llvm_objdump(hipcc=None):
    """llvm-objdump of the ROCm installation the build uses: next to the resolved hipcc (<rocm>/bin/hipcc -> <rocm>/lib/llvm/bin), then
    $ROCM_PATH, /opt/rocm, PATH.  A clear error instead of a FileNotFoundError out of subprocess when none is there."""
    cands = []
    hip = hipcc or os.environ.get("HIPCC") or shutil.which("hipcc")
    if hip:
        hip = shutil.which(hip) or hip
        root = os.path.dirname(os.path.dirname(os.path.realpath(hip)))
        cands += [os.path.join(root, "lib", "llvm", "bin", "llvm-objdump"), os.path.join(root, "llvm", "bin", "llvm-objdump")]
    for root in (os.environ.get("ROCM_PATH"), "/opt/rocm"):
        if root:
            cands.append(os.path.join(root, "lib", "llvm", "bin", "llvm-objdump"))
    cands.append(shutil.which("llvm-objdump"))
    for c in cands:
        if c and os.path.exists(c):
            return c
    raise RuntimeError("isa_check: no llvm-objdump found (looked next to hipcc %r, under $ROCM_PATH, /opt/rocm and on PATH); the ISA invariants of "
                       "the hand-scheduled kernels cannot be verified, so the build stops here" % hip)


def extra_defines():
    """-D switches of AF_HIPCC_EXTRA (the kernel-experiment hook of build.py) as {macro: value}; `-DX`, `-D X`, `-DX=1` all parse."""
    out, toks = {}, shlex.split(os.environ.get("AF_HIPCC_EXTRA", ""))
    i = 0
    while i < len(toks):
        t = toks[i]
        if t == "-D" and i + 1 < len(toks):
            t = "-D" + toks[i + 1]; i += 1
        if t.startswith("-D"):
            k, _, v = t[2:].partition("=")
            out[k.strip()] = v.strip() or "1"
        i += 1
    return out


def dw_slotted_expected():
    """The slotted 8x8 stage is what `k_dw_bf<6>` compiles to unless an experiment switch takes the compiler-scheduled path through the same
    `if constexpr` (dw.hip: DW_SLOT == 0 or any DW_ABL bit)."""
    d = extra_defines()

    def num(v):
        try:
            return int(v, 0)
        except ValueError:
            return 1
    return num(d.get("DW_SLOT", "1")) != 0 and num(d.get("DW_ABL", "0")) == 0


def disassemble(obj):
    """Device ISA of a hipcc object: {kernel symbol: [instruction text, ...]}.  llvm-objdump --offloading writes the bundles next to its
    input, so the object is linked into a temporary directory and unbundled there: nothing in the build directory is touched."""
    objdump = llvm_objdump()
    with tempfile.TemporaryDirectory(prefix="af_isa_") as d:
        base = os.path.basename(obj)
        shutil.copy(os.path.abspath(obj), os.path.join(d, base))
        subprocess.run([objdump, "--offloading", base], cwd=d, check=True, capture_output=True)
        co = [f for f in os.listdir(d) if f.startswith(base + ".") and f.endswith(TARGET)]
        if not co:                        # a unit without device code (host.hip)
            return {}
        assert len(co) == 1, (obj, co)
        txt = subprocess.run([objdump, "-d", os.path.join(d, co[0])], check=True, capture_output=True, text=True).stdout
    kernels, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = kernels.setdefault(m.group(1), [])
            continue
        if cur is not None and line.startswith("\t"):
            cur.append(line.split("//")[0].strip())
    return kernels


def _regs(tok):
    m = re.match(r"([av])\[(\d+):(\d+)\]", tok)
    if m:
        return m.group(1), set(range(int(m.group(2)), int(m.group(3)) + 1))
    m = re.match(r"([av])(\d+)$", tok)
    if m:
        return m.group(1), {int(m.group(2))}
    return None, set()


def check_agpr_fragment_reads(name, ins):
    """(a) ds_read_b128 into AGPRs (the asm fragment reads, invisible to hipcc's s_waitcnt insertion, and the compiler's own): waited for before an
    MFMA reads them.  lgkmcnt(0) retires everything; a COUNTED lgkmcnt(N) (hipcc's own waits in the fp32 blocks) retires all but the N youngest
    LDS operations, which return in order - unless a scalar load is outstanding (SMEM shares the counter and returns out of order): then only
    lgkmcnt(0) counts."""
    queue = []                        # outstanding lgkm operations in issue order: (index, set of AGPRs written, is_smem)
    n = 0
    for i, s in enumerate(ins):
        op = s.split()[0]
        if op.startswith("ds_"):
            regs = set()
            if op == "ds_read_b128" and s.split()[1].startswith("a["):
                _, regs = _regs(s.split()[1].rstrip(","))
                n += 1
            queue.append((i, regs, False))
        elif op.startswith(("s_load", "s_buffer_load")):
            queue.append((i, set(), True))
        elif op == "s_waitcnt" and "lgkmcnt(" in s:
            cnt = int(re.search(r"lgkmcnt\((\d+)\)", s).group(1))
            if cnt == 0:
                queue = []
            elif not any(q[2] for q in queue):
                queue = queue[len(queue) - cnt:] if cnt < len(queue) else queue
        elif op.startswith("v_mfma") and queue:
            toks = [t.strip().rstrip(",") for t in s.split(None, 1)[1].split(",")]
            for t in toks[1:3]:                        # srcA, srcB
                kind, regs = _regs(t)
                if kind != "a":
                    continue
                for qi, qregs, _ in queue:
                    if regs & qregs:
                        raise RuntimeError("%s: %r reads a%d, loaded by the ds_read at instruction %d, which no s_waitcnt in between retires"
                                           % (name, s, min(regs & qregs), qi))
    return n


def check_counted_publish(name, ins, keep=16):
    """(b) the publish's counted vmcnt: exactly `keep` tile stores, and nothing else on the vector-memory counter, behind the last DMA piece."""
    n = 0
    for i, s in enumerate(ins):
        # the publish is the asm `s_waitcnt vmcnt(KEEP) lgkmcnt(0)` directly in front of its s_barrier (hipcc's own counted waits for
        # ordinary loads - e.g. the dPE block's vmcnt(16) - are neither)
        if s.startswith("s_waitcnt") and "vmcnt(%d)" % keep in s and "lgkmcnt(0)" in s and i + 1 < len(ins) and ins[i + 1] == "s_barrier":
            stores = other = 0
            j = i - 1
            while j >= 0 and not ins[j].startswith("global_load_lds"):
                op = ins[j].split()[0]
                if op.startswith("buffer_store_dword") and not op.startswith("buffer_store_dwordx"):
                    stores += 1
                elif op.startswith(("buffer_", "global_", "flat_", "scratch_")):
                    other += 1
                j -= 1
            if j < 0 or stores != keep or other:
                raise RuntimeError("%s: s_waitcnt vmcnt(%d) at instruction %d has %d tile stores and %d other vector-memory instructions behind the last LDS-DMA piece"
                                   % (name, keep, i, stores, other))
            n += 1
    return n


def check_dw_slots(name, ins):
    """(c) the slotted 8x8 loop of k_dw_bf<6>: the branch-free stretch with exactly the 192 MFMAs of two stages."""
    idx = [i for i, s in enumerate(ins) if s.startswith("s_cbranch")]
    best = None
    for a, b in zip(idx, idx[1:]):
        if sum(1 for s in ins[a:b] if s.startswith("v_mfma")) == 192:
            best = (a, b)
    if best is None:
        raise RuntimeError("%s: no branch-free stretch with the 192 MFMAs of two slotted 8x8 stages" % name)
    body = ins[best[0] + 1:best[1]]
    run = worst = 0
    for s in body:
        if s.startswith("v_mfma"):
            run += 1
        elif not s.startswith("s_nop"):
            run = 0
        worst = max(worst, run)
    if worst > 2:
        raise RuntimeError("%s: %d MFMAs back to back in the slotted loop (every MFMA leads its own slot of fillers)" % (name, worst))
    if sum(1 for s in body if s.startswith("s_waitcnt") and "vmcnt(8)" in s) != 2 or sum(1 for s in body if s == "s_barrier") != 2:
        raise RuntimeError("%s: the slotted loop must hold one counted vmcnt(8) wait and one s_barrier per stage" % name)
    dma = [i for i, s in enumerate(body) if s.startswith("global_load_lds_dwordx4")]
    if len(dma) != 16:
        raise RuntimeError("%s: %d LDS-DMA pieces in the slotted loop, expected 16" % (name, len(dma)))
    for i in dma:
        if not (body[i - 1].startswith("s_nop") and body[i - 2].startswith("s_add_u32 m0")) or body[i].rstrip().endswith(" off"):
            raise RuntimeError("%s: LDS-DMA piece %r not in the `s_add_u32 m0 / s_nop / global_load_lds voff, s[base]` form" % (name, body[i - 2:i + 1]))
    fill = [0]
    for s in body:
        if s.startswith("v_mfma"):
            fill.append(0)
        else:
            fill[-1] += 1
    return max(fill[1:-1]), (sum(fill[1:-1]) / float(len(fill) - 2))


def check_valu_write_to_mfma_read(name, ins, need=2):
    """(d) every v_mfma: its SrcA / SrcB (and SrcC, when that is a VGPR) registers were not written by a VALU instruction within the last `need` wait
    states (s_nop N = N + 1 wait states, any other instruction 1).  Checked on the STRAIGHT-LINE instruction order of the kernel: a writer at a loop
    tail feeding an MFMA at the loop head across the back-edge is not followed (the chains' loops start with LDS reads and waits, never with an MFMA),
    and only the first destination of a writer is looked at."""
    n = 0
    for i, s in enumerate(ins):
        if not s.startswith("v_mfma"):
            continue
        ops = [t.strip() for t in s.split(None, 1)[1].split(",")]
        srcs = set()
        for t in ops[1:4]:
            kind, regs = _regs(t.split()[0]) if t else (None, set())
            if kind == "v":
                srcs |= regs
        n += 1
        ws, j = 0, i - 1
        while j >= 0 and ws < need and srcs:
            t = ins[j]
            op = t.split()[0]
            if op == "s_nop":
                ws += int(t.split()[1], 0) + 1
            else:
                if op.startswith("v_") and not op.startswith(("v_mfma", "v_cmp", "v_readlane", "v_readfirstlane")):
                    kind, regs = _regs(t.split(None, 1)[1].split(",")[0].strip()) if len(t.split(None, 1)) > 1 else (None, set())
                    if kind == "v" and regs & srcs:
                        raise RuntimeError("%s: %r reads v%d written by %r only %d wait state(s) earlier (gfx950 needs %d between a VALU write and an MFMA "
                                           "SrcA/SrcB read; an asm VALU op scheduled next to its MFMA consumer?)" % (name, s, min(regs & srcs), t, ws, need))
                ws += 1
            j -= 1
    return n


def check_unit(unit, obj, verbose=True):
    ks = disassemble(obj)
    for name, ins in ks.items():
        if any(s.startswith("scratch_") for s in ins):
            raise RuntimeError("%s: %s uses scratch memory" % (unit, name))
    msg = []
    nm = sum(check_valu_write_to_mfma_read(n, i) for n, i in ks.items())
    if nm:
        msg.append("%d MFMAs, none reads a VGPR a VALU op wrote < 2 wait states earlier" % nm)
    if unit == "mlpbf.hip":
        na = sum(check_agpr_fragment_reads(n, i) for n, i in ks.items())
        nb = sum(check_counted_publish(n, i) for n, i in ks.items())
        if na == 0 or nb == 0:
            raise RuntimeError("mlpbf.hip: the checks found nothing to check (%d AGPR fragment reads, %d counted publishes): the patterns moved" % (na, nb))
        msg.append("%d asm fragment reads waited for, %d counted publishes with exactly 16 stores behind the last DMA piece" % (na, nb))
    if unit == "mlphf.hip":                  # the f16x3 chains: the same two invariants, eight stores behind the last piece (mlphf.hip hf_slot)
        na = sum(check_agpr_fragment_reads(n, i) for n, i in ks.items())
        nb = sum(check_counted_publish(n, i, keep=8) for n, i in ks.items())
        if na == 0 or nb == 0:
            raise RuntimeError("mlphf.hip: the checks found nothing to check (%d AGPR fragment reads, %d counted publishes): the patterns moved" % (na, nb))
        msg.append("%d asm fragment reads waited for, %d counted publishes with exactly 8 stores behind the last DMA piece" % (na, nb))
    if unit == "dw.hip" and dw_slotted_expected():
        k = [n for n in ks if "k_dw_bf" in n and "Li6E" in n]
        assert len(k) == 1, list(ks)
        worst, mean = check_dw_slots(k[0], ks[k[0]])
        msg.append("slotted k_dw_bf<6> loop: 192 MFMAs, at most %d fillers in a slot, %.2f on average" % (worst, mean))
    if verbose and msg:
        print("[isa] %s: %s" % (unit, "; ".join(msg)), flush=True)
    return msg


if __name__ == "__main__":
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for u in sys.argv[1:] or ["mlpbf.hip", "mlphf.hip", "dw.hip", "mlp.hip", "mlp16.hip", "elem.hip"]:
        check_unit(u, os.path.join(here, "build", u.replace(".hip", ".o")))
