import sys,time,os
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import dgtest as T
import discregrid_amd as dg
dg.load_library()
for nu in (71,224):
    V,F=T.icosphere(nu)
    ts=[]
    for _ in range(3):
        t=time.time(); m=dg.Mesh(V,F); ts.append(time.time()-t); del m
    print(nu,len(F),"mesh build (host BVH + upload) min %.3f s"%min(ts), flush=True)
