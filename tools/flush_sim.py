"""Flush-count model of the grad_value sort kernel on the actual base geometry (CPU): taps of every (row, head, level,
point) from the frame plan + the init-grid offsets; per workgroup footprint (128 / 256 / 512 image-ordered rows) the
number of distinct (point group, pixel) runs = memory-side atomics, per level; and the global minimum (distinct pixel
lines per camera and level).  python tools/flush_sim.py > profiles/r4/r4_flush_sim_base.txt"""
import sys, math, numpy as np, torch
sys.path.insert(0,'/root/repo')
from bevformer_amd import synthetic as S
from bevformer_amd.modules import geometry
name='base'; w=S.WORKLOADS[name]
metas=S.make_img_metas(name)
plan=geometry.build_frame_plan(w['bev_h'],w['bev_w'],1,S.PC_RANGE,4,metas,'cpu',row_order='image')
R=plan.row_query.numel(); print('rows',R)
ref=plan.row_ref.numpy()            # (R,4,2)
cam=plan.row_batch.numpy()
shapes=w['shapes']
M=8; P=8
thetas=np.arange(M)*(2*math.pi/M); d=np.stack([np.cos(thetas),np.sin(thetas)],-1); d=d/np.abs(d).max(-1,keepdims=True)
rng=np.random.default_rng(0)
def taps(level, m, rows):
    H,W=shapes[level]
    out=[]
    for p in range(P):
        a=p%4
        x=ref[rows,a,0]*W-0.5 + d[m,0]*(p+1) + rng.normal(0,0.3,len(rows))
        y=ref[rows,a,1]*H-0.5 + d[m,1]*(p+1) + rng.normal(0,0.3,len(rows))
        x0=np.floor(x).astype(int); y0=np.floor(y).astype(int)
        for dx in (0,1):
            for dy in (0,1):
                xx=x0+dx; yy=y0+dy
                ok=(xx>=0)&(yy>=0)&(xx<W)&(yy<H)
                out.append(np.stack([np.full(ok.sum(),a), (yy*W+xx)[ok]],-1))
    return np.concatenate(out,0)
for rpb in (128,256,512):
    tot=np.zeros(4); totg=np.zeros(4); ntaps=np.zeros(4)
    m=1
    for c0 in range(0,R,rpb):
        rows=np.arange(c0,min(R,c0+rpb))
        # split at camera boundary ignored (rare)
        for l in range(4):
            t=taps(l,m,rows)
            ntaps[l]+=len(t)
            totg[l]+=len(np.unique(t[:,0]*10**7+t[:,1]))
            tot[l]+=len(np.unique(t[:,1]))
    print('rows/WG',rpb,'per head: taps',ntaps.astype(int),'flushes (group-keyed)',totg.astype(int),'distinct pixels per WG',tot.astype(int), 'x8 heads total group-keyed', int(totg.sum()*8), 'pixel-keyed', int(tot.sum()*8))
# global distinct per (cam, level)
m=1
for l in range(4):
    g=0
    for c in range(6):
        rows=np.nonzero(cam==c)[0]
        t=taps(l,m,rows); g+=len(np.unique(t[:,1]))
    print('level',l,'global distinct pixel-lines (all cams, one head)',g,'of',6*shapes[l][0]*shapes[l][1])
