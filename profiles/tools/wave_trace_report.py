"""Summary of gpurun_out/trace.npy (wave_trace_run.py): per kernel the wave lifetimes, mean cycles per phase, the
longest-lived waves and one SIMD's timeline."""
import os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t=np.load(os.path.join(ROOT, 'gpurun_out', 'trace.npy'))
for ki,name,marks in ((1,'g2p_grad',['tilefill','sort','adj+barrier','loop','flush','Xarrived']),(2,'p2g_grad',['-','fill','compute','store']),(0,'g2p_p2g',['fill issued','sort+E+barrier','gather+store','bbox2+zero','const+scatter','flush'])):
    a=t[ki][:7944].astype(np.int64)
    a=a[a[:,0]>0]
    t0=a[:,0].min()
    st=a[:,0]-t0
    end=a[:,1:11].max(1)-t0
    print(name,'waves',len(a),'kernel span cycles',end.max(), '= us@2.4GHz', round(end.max()/2400,1))
    life=end-st
    rows=np.nonzero(t[ki][:7944,0]>0)[0]
    top=np.argsort(-life)[:12]
    print('  longest waves (workgroup, start, lifetime):', [(int(rows[i]//4), int(st[i]), int(life[i])) for i in top])
    print('  lifetime pct 10/50/90', np.percentile(life,[10,50,90]).astype(int), ' start-time pct 25/50/75/100', np.percentile(st,[25,50,75,100]).astype(int))
    # mean phase durations in order of marks: for g2p_grad order: start->X(5)->tilefill(0)->sort(1)->adj(2)->loop(3)->flush(4)
    if ki==1: order=[6,1,2,3,4,5]
    elif ki==2: order=[2,3,4]
    else: order=[1,2,3,4,5,6]
    prev=a[:,0]
    for o in order:
        d=a[:,o]-prev; prev=a[:,o]
        print(f'   -> mark {o-1} {marks[o-1]:18s} mean {d.mean():9.0f} p50 {np.median(d):8.0f} p90 {np.percentile(d,90):8.0f}')
    hw=a[:,11]; cu=(hw>>8)&0xf; sh=(hw>>12)&1; se=(hw>>13)&0x7; simd=(hw>>4)&3
    key=a[:,12]*1000+se*100+sh*16+cu
    u,c=np.unique(key,return_counts=True)
    k0=u[len(u)//3]
    sel=np.where((key==k0)&(simd==0))[0]; sel=sel[np.argsort(st[sel])]
    print('  one SIMD timeline (start,end):', [(int(st[i]),int(end[i])) for i in sel])
