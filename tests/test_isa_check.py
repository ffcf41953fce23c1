"""The build-time ISA checks (all-in-one-deflicker_amd/isa_check.py): they must pass on the objects of the current
build and FAIL on instruction streams that break the invariants the hand-scheduled kernels rely on (CPU only: the checker reads
disassembly, hipcc cross-compiles)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "all-in-one-deflicker_amd")


def _mod():
    spec = importlib.util.spec_from_file_location("af_isa_check", os.path.join(PKG, "isa_check.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_fragment_read_without_wait_is_caught():
    m = _mod()
    ok = ["ds_read_b128 a[0:3], v4 offset:512", "v_add_f32_e32 v1, v2, v3", "s_waitcnt lgkmcnt(0)", "v_mfma_f32_32x32x16_bf16 a[16:31], a[0:3], v[8:11], a[16:31]"]
    assert m.check_agpr_fragment_reads("k", ok) == 1
    bad = [ok[0], ok[1], ok[3]]
    with pytest.raises(RuntimeError, match="no s_waitcnt in between retires"):
        m.check_agpr_fragment_reads("k", bad)
    # a wait that only counts the vector-memory queue does not cover the LDS read
    with pytest.raises(RuntimeError):
        m.check_agpr_fragment_reads("k", [ok[0], "s_waitcnt vmcnt(0)", ok[3]])
    # a COUNTED wait (hipcc's own, in the fp32 blocks) retires all but the N youngest LDS operations: the first of two reads is covered by lgkmcnt(1),
    # the second is not; with a scalar load outstanding (out-of-order returns on the same counter) only lgkmcnt(0) counts
    two = [ok[0], "ds_read_b128 a[4:7], v4 offset:1024", "s_waitcnt lgkmcnt(1)"]
    assert m.check_agpr_fragment_reads("k", two + [ok[3]]) == 2
    with pytest.raises(RuntimeError):
        m.check_agpr_fragment_reads("k", two + ["v_mfma_f32_32x32x16_bf16 a[16:31], a[4:7], v[8:11], a[16:31]"])
    with pytest.raises(RuntimeError):
        m.check_agpr_fragment_reads("k", ["s_load_dwordx2 s[0:1], s[4:5], 0x0"] + two + [ok[3]])
    # an MFMA on other registers may issue while the read is in flight
    assert m.check_agpr_fragment_reads("k", [ok[0], "v_mfma_f32_32x32x16_bf16 a[16:31], a[4:7], v[8:11], a[16:31]", ok[2], ok[3]]) == 1


def test_valu_write_next_to_its_mfma_reader_is_caught():
    """Rule (d), the pair behind round 3's corrupted two-layer chains: gfx950 needs two wait states between a VALU write of a VGPR and an MFMA
    reading it as SrcA / SrcB (tools/hazardprobe.hip on the MI355X: 0 or 1 wait states -> the OLD register content in > 96 % of the lanes,
    2 -> never).  hipcc pads its own VALU instructions and cannot see one inside an asm statement: the instruction streams below are the
    shapes the unfenced build (-DAF_NO_ELEMWISE_FENCE) produced and the shapes that are safe."""
    m = _mod()
    relu, mfma = "v_max_f32_e32 v10, 0, v2", "v_mfma_f32_4x4x1_16b_f32 a[0:3], v9, v10, a[0:3]"
    with pytest.raises(RuntimeError, match="only 0 wait state"):
        m.check_valu_write_to_mfma_read("k", [relu, mfma])
    with pytest.raises(RuntimeError, match="only 1 wait state"):                           # the five pairs of the unfenced mlpbf.o: one instruction in between
        m.check_valu_write_to_mfma_read("k", [relu, "v_accvgpr_read_b32 v3, a17", mfma])
    with pytest.raises(RuntimeError, match="only 1 wait state"):
        m.check_valu_write_to_mfma_read("k", [relu, "s_nop 0", mfma])
    with pytest.raises(RuntimeError):                                                      # SrcA as well, and a write of part of a register tuple
        m.check_valu_write_to_mfma_read("k", ["v_cvt_pk_bf16_f32 v5, v20, v21", "s_nop 0", "v_mfma_f32_32x32x16_bf16 a[0:15], v[4:7], v[8:11], a[0:15]"])
    assert m.check_valu_write_to_mfma_read("k", [relu, "s_nop 1", mfma]) == 1              # two wait states: safe
    assert m.check_valu_write_to_mfma_read("k", [relu, "v_accvgpr_read_b32 v3, a17", "v_max_f32_e32 v11, 0, v3", mfma]) == 1
    assert m.check_valu_write_to_mfma_read("k", ["v_max_f32_e32 v12, 0, v2", mfma]) == 1   # another register
    assert m.check_valu_write_to_mfma_read("k", ["v_accvgpr_write_b32 a9, v2", "v_mfma_f32_32x32x16_bf16 a[0:15], a[8:11], v[8:11], a[0:15]"]) == 1   # AGPR sources are another hazard class (hipcc's own instructions)


@pytest.mark.skipif(not os.environ.get("AF_SLOW_ISA_TEST"), reason="compiles mlpbf.hip without the element-wise fence (~1 min): AF_SLOW_ISA_TEST=1")
def test_the_unfenced_build_is_rejected(tmp_path):
    """The real thing: mlpbf.hip compiled with -DAF_NO_ELEMWISE_FENCE puts an asm v_max_f32 one wait state in front of the v_mfma_f32_4x4x1
    that reads it (and fails tests/test_gpu_arch.py on the MI355X: gpurun_out r5b/pytest_nofence.log); rule (d) must refuse it."""
    import subprocess
    m = _mod()
    obj = str(tmp_path / "mlpbf_nofence.o")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DAF_NO_ELEMWISE_FENCE", "-c", os.path.join(PKG, "csrc", "mlpbf.hip"), "-o", obj], check=True)
    with pytest.raises(RuntimeError, match="wait state"):
        m.check_unit("mlpbf.hip", obj, verbose=False)


def test_counted_publish_is_checked():
    m = _mod()
    stores = ["buffer_store_dword v%d, v1, s[4:7], 0 offen" % i for i in range(16)]
    ok = ["global_load_lds_dwordx4 v[2:3], off"] + stores[:5] + ["v_mfma_f32_32x32x16_bf16 a[0:15], a[16:19], v[8:11], a[0:15]"] + stores[5:] + ["s_waitcnt vmcnt(16) lgkmcnt(0)", "s_barrier"]
    assert m.check_counted_publish("k", ok) == 1
    assert m.check_counted_publish("k", ok[:-2] + ["s_waitcnt vmcnt(16)", "v_add_f32_e32 v1, v2, v3"]) == 0      # hipcc's own counted wait for ordinary loads: not a publish
    with pytest.raises(RuntimeError, match="15 tile stores"):          # a store merged away / moved in front of the piece
        m.check_counted_publish("k", ok[:3] + ok[4:])
    with pytest.raises(RuntimeError, match="1 other"):                 # any other vector-memory instruction breaks the count
        m.check_counted_publish("k", ok[:-2] + ["global_load_dword v9, v[2:3], off"] + ok[-2:])
    with pytest.raises(RuntimeError):                                  # two stores merged into one wide store
        m.check_counted_publish("k", ok[:1] + ["buffer_store_dwordx2 v[0:1], v1, s[4:7], 0 offen"] + stores[2:] + ok[-2:])


def _slot_loop(m_per_gap=4, clump=False):
    body = []
    dma = 0
    for i in range(192):
        body.append("v_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]")
        if clump and i % 2 == 0:
            continue
        if i in (0, 96):
            body += ["s_waitcnt vmcnt(8)", "s_barrier"]
        if i % 12 == 7:
            body += ["s_add_u32 m0, s4, 0x1000", "s_nop 0", "global_load_lds_dwordx4 v9, s[2:3]"]; dma += 1
        body += ["v_sub_f32_e32 v1, v2, v3"] * m_per_gap
    assert clump or dma == 16
    return ["s_cbranch_scc1 12"] + body + ["s_cbranch_scc0 65000", "s_endpgm"]


def test_slotted_loop_shape_is_checked():
    m = _mod()
    worst, mean = m.check_dw_slots("k", _slot_loop())
    assert worst <= 9 and 4.0 <= mean <= 5.0
    bad = _slot_loop()
    bad[5:5] = ["v_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]"]           # 193 MFMAs: not the two stages
    with pytest.raises(RuntimeError, match="192 MFMAs"):
        m.check_dw_slots("k", bad)
    run = ["s_cbranch_scc1 12"] + ["v_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]"] * 192 + ["s_cbranch_scc0 1"]
    with pytest.raises(RuntimeError, match="back to back"):                             # the MFMAs sunk below their fillers (what hipcc did before the results were tied to their slots)
        m.check_dw_slots("k", run)
    off = [s.replace("global_load_lds_dwordx4 v9, s[2:3]", "global_load_lds_dwordx4 v[8:9], off") for s in _slot_loop()]
    with pytest.raises(RuntimeError, match="form"):                                     # per-lane 64-bit address instead of s[base] + voff
        m.check_dw_slots("k", off)


@pytest.mark.skipif(not os.path.exists(os.path.join(PKG, "build", "mlpbf.o")), reason="objects not built here")
def test_current_objects_pass():
    m = _mod()
    for unit in ("mlpbf.hip", "dw.hip"):
        msg = m.check_unit(unit, os.path.join(PKG, "build", unit.replace(".hip", ".o")), verbose=False)
        assert msg, unit
