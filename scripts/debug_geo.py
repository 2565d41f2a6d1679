import ctypes, numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roitr_amd import _lib as L
s = np.load('tests/golden/stages.npz')
pts = torch.from_numpy(s['geo.points'][0]).cuda().contiguous()
n = pts.shape[0]
off = torch.tensor([n], dtype=torch.int32).cuda()
con = torch.zeros(n, dtype=torch.int32).cuda()
eoff = torch.zeros(1, dtype=torch.int64).cuda()
d = torch.zeros(n*n).cuda(); a = torch.zeros(n*n*3).cuda()
lib = L.lib()
L.check(lib.roitr_geo_indices(n, L.ptr(pts), L.ptr(off), L.ptr(con), L.ptr(eoff), ctypes.c_float(0.2), ctypes.c_float(15.0), 3, n, L.ptr(d), L.ptr(a), L.stream_ptr()))
torch.cuda.synchronize()
dd = d.cpu().numpy().reshape(n,n); aa = a.cpu().numpy().reshape(n,n,3)
print('d_idx err', np.abs(dd - s['geo.d_idx'][0]).max(), 'a_idx err', np.abs(aa - s['geo.a_idx'][0]).max())
bad = np.argwhere(np.abs(aa - s['geo.a_idx'][0]) > 1e-3)
print(bad[:10]); 
if len(bad): 
    i,j,k = bad[0]; print(aa[i,j], s['geo.a_idx'][0][i,j])
# sinusoid
from roitr_amd.riga import div_term
dv = div_term(256).cuda()
S = torch.zeros(n*n, 256).cuda()
lib.roitr_sinusoid.argtypes = None
L.check(lib.roitr_sinusoid(ctypes.c_long(n*n), 256, L.ptr(d), L.ptr(dv), L.ptr(S), L.stream_ptr()))
torch.cuda.synchronize()
ref = torch.from_numpy(s['geo.d_idx'][0]).reshape(-1,1,1) * div_term(256).view(1,-1,1)
refS = torch.cat([torch.sin(ref), torch.cos(ref)], 2).view(n*n, 256)
print('sinusoid err', (S.cpu()-refS).abs().max().item())
