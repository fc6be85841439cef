"""Design study (CPU emulator): node steps of the filtered K1 traversal by depth of the node in the tree -- what a start list of
subtree roots per coarse cell could save at best.  usage: python tests/perf/emu_depth_study.py [ico|bunny] [res]"""
import sys, os
os.environ["EMU_DEPTH_HIST"]="1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, dgtest as T, emu, ctypes as C
mesh=sys.argv[1] if len(sys.argv)>1 else 'ico'
res=int(sys.argv[2]) if len(sys.argv)>2 else 256
V,F=(T.icosphere(71) if mesh=='ico' else T.bunny_mesh())
dom=T.oracle_default_domain(V); R=[res]*3
m=emu.EmuMesh(V,F)
plane=(res+1)**2
runs=[(int(k)*4*plane,(int(k)*4+4)*plane) for k in np.linspace(0,(res+1)//4-1,8)]
emu.set_fast(1)
for b,e in runs: m.sample_range(dom,R,b,e)
fs=emu.fast_stats()
h=np.zeros(64,dtype=np.uint64); emu.lib().emu_depth_hist(h.ctypes.data_as(C.c_void_p))
B=fs['bricks']; print(mesh,res,'bricks',B,'pair steps/brick',fs['pair_steps']/B,'leaf visits',fs['leaf_visits']/B)
cum=0
for d in range(40):
    if h[d]: cum+=h[d]/B; print('depth %2d: %.2f steps/brick  (cumulative %.2f)'%(d,h[d]/B,cum))
