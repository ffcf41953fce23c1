"""Numerical study (numpy, CPU): error of a K = 256 dot product against fp64 for the fp32 fmaf chain, the bf16x6 scheme the
kernels use, bf16x3, and a two-term fp16 split with a per-tensor power-of-two scale (3 or 4 products) — the candidate for
halving the matrix-pipe cycles again.  Result: with benign value distributions fp16x3 is as accurate as
bf16x6 (rms 9e-9 of |a||b|), but a tensor whose rows span more than ~2^28 in magnitude loses its small rows completely
(fp16 has 5 exponent bits): it needs a range guarantee or per-row scales before it may replace bf16x6.
Usage: python tools/experiments/split_error_study.py"""
import numpy as np
rng=np.random.default_rng(0)
def bf16(x):
    x=np.asarray(x,np.float32); u=x.view(np.uint32).astype(np.uint64)
    r=((u+0x7fff+((u>>16)&1))&0xffff0000).astype(np.uint32)
    return r.view(np.float32)
def split_bf3(x):
    h=bf16(x); r1=(x-h).astype(np.float32); m=bf16(r1); r2=(r1-m).astype(np.float32); l=bf16(r2); return h,m,l
def split_f16(x,s):
    xs=(x*np.float32(s)).astype(np.float32)
    h=xs.astype(np.float16).astype(np.float32); r=(xs-h).astype(np.float32); l=r.astype(np.float16).astype(np.float32)
    return h,l
def dot_f32acc(terms):   # terms: list of (a,b) arrays [M,K]; products exact in fp64 then accumulate sequentially in fp32 per k (emulating MFMA fp32 accumulate, 16-wide groups summed exactly)
    M,K=terms[0][0].shape
    acc=np.zeros(M,np.float32)
    for k0 in range(0,K,16):
        for a,b in terms:
            p=(a[:,k0:k0+16].astype(np.float64)*b[:,k0:k0+16].astype(np.float64)).sum(1)
            acc=(acc.astype(np.float64)+p).astype(np.float32)
    return acc
def fmaf_chain(a,b):
    acc=np.zeros(a.shape[0],np.float32)
    for k in range(a.shape[1]):
        acc=(acc.astype(np.float64)+a[:,k].astype(np.float64)*b[:,k].astype(np.float64)).astype(np.float32)
    return acc
def run(name,a,b):
    ref=(a.astype(np.float64)*b.astype(np.float64)).sum(1)
    scale=np.sqrt((a.astype(np.float64)**2).sum(1)*(b.astype(np.float64)**2).sum(1))   # norm-wise
    e={}
    e['fp32 fmaf']=fmaf_chain(a,b)
    ah,am,al=split_bf3(a); bh,bm,bl=split_bf3(b)
    e['bf16x6']=dot_f32acc([(al,bh),(ah,bl),(am,bm),(am,bh),(ah,bm),(ah,bh)])
    e['bf16x3']=dot_f32acc([(am,bh),(ah,bm),(ah,bh)])
    sa=2.0**(14-np.ceil(np.log2(np.abs(a).max()))); sb=2.0**(14-np.ceil(np.log2(np.abs(b).max())))
    ah,al=split_f16(a,sa); bh,bl=split_f16(b,sb)
    e['fp16x3 (hh,hl,lh; per-tensor 2^k scale)']=dot_f32acc([(al,bh),(ah,bl),(ah,bh)])/np.float32(sa*sb)
    e['fp16x4 (+ll)']=dot_f32acc([(al,bl),(al,bh),(ah,bl),(ah,bh)])/np.float32(sa*sb)
    print(name)
    for k,v in e.items():
        err=np.abs(v.astype(np.float64)-ref)
        print('   %-44s rms err / (|a||b|) = %.3e   max = %.3e   rms rel to |result| = %.3e'%(k, np.sqrt(np.mean((err/scale)**2)), (err/scale).max(), np.sqrt(np.mean((err/np.abs(ref).clip(1e-30))**2))))
M,K=4000,256
run('uniform(-1,1) x uniform(-1,1), K=256', rng.uniform(-1,1,(M,K)).astype(np.float32), rng.uniform(-1,1,(M,K)).astype(np.float32))
# activations: post-ReLU (half zeros), weights ~ U(-1/16,1/16)
act=np.maximum(rng.normal(0,1,(M,K)),0).astype(np.float32); w=rng.uniform(-1/16,1/16,(M,K)).astype(np.float32)
run('relu activations x kaiming weights, K=256', act, w)
# gradients: heavy-tailed small values (lognormal magnitudes spanning 2^20) x activations
gz=(rng.normal(0,1,(M,K))*np.exp(rng.normal(0,3.0,(M,K)))*1e-6).astype(np.float32)
run('heavy-tailed gradients (lognormal sigma 3, ~1e-6) x relu activations, K=256', gz, act)
gz2=(rng.normal(0,1,(M,K))*np.exp(rng.normal(0,6.0,(M,K)))*1e-8).astype(np.float32)
run('very heavy-tailed gradients (lognormal sigma 6) x relu activations, K=256', gz2, act)
