"""Dev probe (CPU only): how often would the "dive" of nastar_search_asm3.hip.h hit on the LONGEST searches of the bench batches (the ones a
launch waits for)?  A numpy re-walk of the reference state machine counts the steps whose selected node was relaxed in the previous
step (exact) and those that also pass the kernel's conservative test (its key strictly below / tie-broken below the previous minimum).
Round 3: maze32 26-28 %, rand32 50-73 %, rand64 67-78 %; the conservative test loses ~1 %.   Usage: python tools/sim_dive.py"""
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'neural-astar_amd')]
import bench
from oracle import oracle as O
def walk(cost, passable, s, go, H, W, gr=0.5):
    sq=np.float32(np.sqrt(np.float32(W))); grf=np.float32(gr); omg=np.float32(1-gr)
    rr,cc=np.meshgrid(np.arange(H,dtype=np.float32),np.arange(W,dtype=np.float32),indexing="ij")
    dr=np.abs(rr-np.float32(go//W)); dc=np.abs(cc-np.float32(go%W))
    h0=((dr+dc)-np.minimum(dr,dc))+np.float32(0.001)*np.sqrt(dr*dr+dc*dc)
    c=cost.reshape(-1).astype(np.float32); p=passable.reshape(-1)>0
    h=(h0.reshape(-1).astype(np.float32)+c).astype(np.float32)
    g=np.zeros(H*W,np.float32); opn=np.zeros(H*W,bool); clo=np.zeros(H*W,bool); opn[s]=True
    hits=hits_exact=steps=0; prevU=None; prevM=None; prevS=None
    while True:
        f=(grf*g).astype(np.float32)+(omg*h).astype(np.float32); q=(f/sq).astype(np.float32)
        idx=np.nonzero(opn)[0]
        if idx.size==0: break
        k=int(np.argmin(q[idx])); sel=int(idx[k]); M=q[sel]
        if prevU is not None:
            if sel in prevU:
                hits_exact+=1
                if (M<prevM) or (M==prevM and sel<prevS): hits+=1
        steps+=1
        clo[sel]=True
        if sel==go: break
        opn[sel]=False
        r0,c0=divmod(sel,W); g2=np.float32(g[sel]+c[sel]); U=set()
        for a in (-1,0,1):
            for b in (-1,0,1):
                if a==0 and b==0: continue
                r1,c1=r0+a,c0+b
                if 0<=r1<H and 0<=c1<W:
                    n=r1*W+c1
                    if p[n] and ((not opn[n] and not clo[n]) or (opn[n] and g[n]>g2)):
                        g[n]=g2; opn[n]=True; U.add(n)
        prevU,prevM,prevS=U,M,sel
    return steps,hits,hits_exact
for kind,HW in (("maze32",32),("rand32",32),("rand64",64)):
    pr=bench.make_problem(kind,4096,1234)
    it=O.forward(pr.map_designs,pr.start_maps,pr.goal_maps,pr.map_designs,0.5,HW*HW,mode="sm").iters
    order=np.argsort(-it)[:3]
    for b in order:
        m=pr.map_designs[b,0]; s=int(pr.start_maps[b].reshape(-1).argmax()); go=int(pr.goal_maps[b].reshape(-1).argmax())
        st,h,he=walk(m,m,s,go,HW,HW)
        print(kind,"map",b,"iters",it[b],"steps",st,"conservative hits",h,round(h/st,2),"exact-next-in-U",he,round(he/st,2))
