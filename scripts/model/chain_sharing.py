"""CPU model (tuning aid, no GPU): how much of cross-based aggregation's work is REDUNDANT on a pair with real-scene arms (round 5)?
The reference's sum of an output is one chain of additions, rows ascending; outputs of one column whose supports start at the same row share that chain
as a prefix (and are identical where they also end at the same row).  Per column and 16-row step of the tile kernel: the number of distinct chains
among the 16 outputs, and the taps a one-accumulator walk per chain would add against the taps the kernel's four-accumulator item walk touches.
    python scripts/model/chain_sharing.py [natural|sample]"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
from oracle import cpu_oracle as oracle
from util import natural_pair, sample_pair
H,W,D,L1,tau1 = 1000,1500,256,14,0.02
which = sys.argv[1] if len(sys.argv) > 1 else 'natural'
x0,x1 = natural_pair(H,W,D,seed=1234) if which=='natural' else sample_pair(H,W)
a0 = np.asarray(oracle.cross(x0,L1,tau1)).reshape(4,H,W).astype(np.int32)
a1 = np.asarray(oracle.cross(x1,L1,tau1)).reshape(4,H,W).astype(np.int32)
def decode(a):
    ys,xs = np.mgrid[0:H,0:W]
    return xs-a[0]-1, a[1]-xs-1, ys-a[2]-1, a[3]-ys-1
l0,r0,u0,d0 = decode(a0); l1,r1,u1,d1 = decode(a1)
ys = np.arange(H)[:,None]
for d in [7,60,130,200]:
    sh=-d
    xs=np.arange(W); ok=(xs+sh>=0)&(xs+sh<W); xp=np.clip(xs+sh,0,W-1)
    l=np.minimum(l0,l1[:,xp]); r=np.minimum(r0,r1[:,xp]); u=np.minimum(u0,u1[:,xp]); dn=np.minimum(d0,d1[:,xp])
    n=l+r+1
    first = ys-u; last = ys+dn
    # prefix sums of n over rows per column
    cs = np.concatenate([np.zeros((1,W),np.int64), np.cumsum(n,0)],0)
    taps = cs[last+1, xs[None,:]] - cs[first, xs[None,:]]   # additions per output
    okm = np.broadcast_to(ok[None,:], (H,W))
    print('d',d,'adds/voxel %.1f'%taps[okm].mean(), 'nonmin %.3f'%((taps[okm]!=9).mean()))
    # chains: per column per 16-row step: distinct first rows; chain taps = cs[maxlast+1]-cs[first]
    TH=16
    tot_chain_taps=0; tot_out_taps=0; nchains=0; nsteps=0; ndistinct=0
    cur_item_taps=0
    hist_nch=np.zeros(17,int)
    chain_taps_by_len = {}
    for ty in range(0,H-TH+1,TH):
        F = first[ty:ty+TH]; L = last[ty:ty+TH]
        # for each column: unique firsts
        for c in range(0,W,7):   # sample columns
            if not ok[c]: continue
            f=F[:,c]; la=L[:,c]
            uf=np.unique(f)
            hist_nch[len(uf)]+=1
            nchains+=len(uf); nsteps+=1
            ndistinct += len(set(zip(f.tolist(),la.tolist())))
            for ff in uf:
                ml = la[f==ff].max()
                t = cs[ml+1,c]-cs[ff,c]
                tot_chain_taps += t
            tot_out_taps += taps[ty:ty+TH,c].sum()
            # current: items of 4 rows: walk top..bot rows: taps walked
            for g in range(4):
                tp=f[4*g:4*g+4].min(); bt=la[4*g:4*g+4].max()
                cur_item_taps += cs[bt+1,c]-cs[tp,c]
    print('  per column-step: chains %.2f distinct outputs %.2f of 16; chain taps/ out taps %.3f ; item-walk taps(x1)/out taps %.3f'%(nchains/nsteps, ndistinct/nsteps, tot_chain_taps/tot_out_taps, cur_item_taps/tot_out_taps))
    print('  nchains hist', hist_nch.tolist())
    # instr estimate per tap: current 5 per item-walk tap (cmpx+4add) ; chain scheme 2 per chain tap
    print('  tap instr: current %.2f per out-tap, chain(1acc) %.2f per out-tap'%(5*cur_item_taps/tot_out_taps, 2*tot_chain_taps/tot_out_taps))
