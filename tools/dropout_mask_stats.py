import numpy as np
M32=np.uint64(0xFFFFFFFF)
def mix32(x):
    x=x.astype(np.uint64); x^=x>>np.uint64(16); x=(x*np.uint64(0x85EBCA6B))&M32; x^=x>>np.uint64(13); x=(x*np.uint64(0xC2B2AE35))&M32; x^=x>>np.uint64(16); return x
def rowkey(seed,rows):
    rows=rows.astype(np.uint64)
    return mix32(((rows&M32)*np.uint64(0x9E3779B1) + mix32(np.full(rows.shape, seed, np.uint64)))&M32)
def h_old(rk,pair):
    x=rk[:,None]^((pair[None,:].astype(np.uint64)*np.uint64(0x9E3779B1))&M32)
    x^=x>>np.uint64(15); x=(x*np.uint64(0x2C1B3C6D))&M32; x^=x>>np.uint64(13); return x
def h_new(rk,pair,C=0x2C1B3C6D):
    x=rk[:,None]^((pair[None,:].astype(np.uint64)*np.uint64(0x9E3779B1))&M32)
    return (x*np.uint64(C))>>np.uint64(32)
def flags(h,th):
    lo=(h&np.uint64(0xFFFF))>=th; hi=(h>>np.uint64(16))>=th
    out=np.empty((h.shape[0],h.shape[1]*2),bool); out[:,0::2]=lo; out[:,1::2]=hi; return out
def stats(name,f,p):
    n=f.size; keep=f.mean()
    rowm=f.mean(1); colm=f.mean(0)
    # binomial expectations
    sr=np.sqrt(p*(1-p)/f.shape[1]); sc=np.sqrt(p*(1-p)/f.shape[0])
    d=1.0-f  # drop indicator
    def corr(a,b):
        a=a-a.mean(); b=b-b.mean(); return float((a*b).mean()/np.sqrt((a*a).mean()*(b*b).mean()))
    c_k1=corr(d[:,:-1],d[:,1:]); c_k2=corr(d[:,:-2],d[:,2:]); c_q1=corr(d[:-1],d[1:]); c_q2=corr(d[:-2],d[2:])
    c_diag=corr(d[:-1,:-1],d[1:,1:])
    print(f"{name}: drop rate {1-keep:.5f} (p {p}); row-rate std {rowm.std():.5f} (binomial {sr:.5f}); col-rate std {colm.std():.5f} (binomial {sc:.5f}); "
          f"corr key+1 {c_k1:+.4f} key+2 {c_k2:+.4f} query+1 {c_q1:+.4f} query+2 {c_q2:+.4f} diag {c_diag:+.4f}  (noise ~{1/np.sqrt(n):.4f})")
for p in (0.1,0.5):
    th=np.uint64(int(p*65536+0.5))
    for seed in (1,12345,0x7FFFFFFF):
        rows=np.arange(3072*196//16,dtype=np.uint64)+np.uint64(777)    # consecutive rows as the kernels use them
        rk=rowkey(seed,rows)
        pair=np.arange(1536,dtype=np.uint64)    # 3072 columns (GEMM width) -- attention uses the first 98
        for nm,fn in (("old",h_old),("new",h_new)):
            f=flags(fn(rk,pair),th)
            stats(f"p={p} seed={seed} {nm} [wide]",f,p)
            stats(f"p={p} seed={seed} {nm} [196 cols]",f[:,:196],p)
print("=========== per-element multiply-shift: keep iff mul_lo(rk ^ elem*G, C) >= th << 16")
def f_elem(rk,ncols,th,C=0x2C1B3C6D,G=0x9E3779B1,pre=False):
    e=np.arange(ncols,dtype=np.uint64)
    x=rk[:,None]^((e[None,:]*np.uint64(G))&M32)
    if pre: x^=x>>np.uint64(15)
    h=(x*np.uint64(C))&M32
    return h>=(th<<np.uint64(16))
for p in (0.1,0.5):
    th=np.uint64(int(p*65536+0.5))
    for seed in (1,12345):
        rows=np.arange(3072*196//16,dtype=np.uint64)+np.uint64(777)
        rk=rowkey(seed,rows)
        for C in (0x2C1B3C6D,0x85EBCA6B,0xC2B2AE35):
            f=f_elem(rk,3072,th,C)
            stats(f"p={p} seed={seed} elem C={C:08X} [wide]",f,p)
            stats(f"p={p} seed={seed} elem C={C:08X} [196]",f[:,:196],p)
        f=f_elem(rk,3072,th,pre=True)
        stats(f"p={p} seed={seed} elem+preshift [wide]",f,p)
# also: consecutive seeds (sites of one step differ by the seed only): cross-site correlation
th=np.uint64(int(0.1*65536+0.5)); rows=np.arange(20000,dtype=np.uint64)
a=f_elem(rowkey(100,rows),196,th); b=f_elem(rowkey(101,rows),196,th)
d1=1.0-a; d2=1.0-b
print("cross-seed correlation (seed 100 vs 101):", float(((d1-d1.mean())*(d2-d2.mean())).mean()/d1.std()/d2.std()))
