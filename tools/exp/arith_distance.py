"""Pose distance between the two arithmetic sets of the CPU oracle (OpenCV's generic paths / legacy), per schedule, size and depth mode.
Test infrastructure: runs oracle/ only.  usage: python tools/exp/arith_distance.py"""
import sys, importlib, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O
synth = importlib.import_module("uw-slam_amd.synth")
def rot(qa,qb):
    qa,qb=qa.astype(float),qb.astype(float)
    w=abs(np.dot(qa,qb)); v=qb[3]*qa[:3]-qa[3]*qb[:3]-np.cross(qa[:3],qb[:3])
    return 2*np.arctan2(np.linalg.norm(v),w)
for (w,h,f,cx,cy) in ((320,240,262.5,159.5,119.5),(640,480,525.0,319.5,239.5)):
  for name,over in (("fixed4x10",dict(n_levels=4, first_level=3, last_level=0, max_iters=10, early_exit=0)),("reference",dict())):
    for depth in (False,True):
      dts=[];dqs=[];its=0
      for s in range(4000,4000+(8 if w==640 else 16)):
        ref,tgt,dep,_,_=synth.render_pair(w,h,f,f,cx,cy,seed=s,with_depth=depth)
        r=[]
        for ar in (0,1):
            p=O.default_params(w,h,f,f,cx,cy,arith=ar,**over)
            if depth: p.has_depth=1
            st,pose,tr=O.align_pair(p,ref,tgt,dep if depth else None,want_trace=True)
            r.append((pose,len(tr)))
        dts.append(float(np.linalg.norm(r[0][0][4:].astype(float)-r[1][0][4:].astype(float))))
        dqs.append(rot(r[0][0][:4],r[1][0][:4]))
        its+= r[0][1]!=r[1][1]
      print(w,h,name,"depth" if depth else "z=1","max|dt| %.3e median %.3e  max dq %.3e  iter-count differs %d/%d"%(max(dts),np.median(dts),max(dqs),its,len(dts)))
