import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import torch, numpy as np
from xgcm_amd import DataArray, device as D
from bench_configs import mitgcm_grid, timeit
nz,ny,nx=75,2400,3600
grid=mitgcm_grid(nz,ny,nx)
T=DataArray(D.synthetic((nz,ny,nx),2),("Z","YC","XC"))
dx=grid._ds["dxC"].data
print("dxC", tuple(dx.shape), dx.stride(), dx.is_contiguous(), dx.data_ptr()%16)
for _ in range(2):
    print("grid.derivative X", round(timeit(lambda: grid.derivative(T,"X"), 7),3))
    print("raw stencil m_out=dx[None]", round(timeit(lambda: D.stencil1d("diff",T.data,2,1,0,"periodic",m_out=dx[None]),7),3))
    d2=D.synthetic((1,ny,nx),31,0,1000.0,1000.0)
    print("raw stencil m_out=fresh (1,ny,nx)", round(timeit(lambda: D.stencil1d("diff",T.data,2,1,0,"periodic",m_out=d2),7),3))
    print("grid.diff X", round(timeit(lambda: grid.diff(T,"X"), 7),3))
m=grid.get_metric(type("X",(),{"dims":("Z","YC","XG"),"name":None})(), ("X",))
print("metric picked:", m.dims, tuple(m.shape))
