"""Host-side cost of one `Grid` call (Python dispatch + ctypes + launch) on tiny HBM-resident arrays.

    python tools/hostprof.py        # on a GPU box; measured 40-60 us per operator, 12 us for the raw ABI call
"""
import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from xgcm_amd import DataArray, Dataset, Grid
from xgcm_amd import device as D
nz,ny,nx=2,4,128
coords={"XC":("XC",np.arange(nx)+.5),"XG":("XG",np.arange(nx)*1.),"YC":("YC",np.arange(ny)+.5),"YG":("YG",np.arange(ny)*1.),"Z":("Z",np.arange(nz)+.5),"Zl":("Zl",np.arange(nz)*1.)}
ds=Dataset({"dxC":DataArray(D.synthetic((ny,nx),1,0,1.,1.),("YC","XG")),"dyC":DataArray(D.synthetic((ny,nx),2,0,1.,1.),("YG","XC")),"drF":DataArray(D.synthetic((nz,),3,0,1.,1.),("Z",))},coords)
g=Grid(ds,coords={"X":{"center":"XC","left":"XG"},"Y":{"center":"YC","left":"YG"},"Z":{"center":"Z","left":"Zl"}},padding={"X":"periodic","Y":"extend","Z":"fill"},metrics={("X",):["dxC"],("Y",):["dyC"],("Z",):["drF"]},autoparse_metadata=False)
T=DataArray(D.synthetic((nz,ny,nx),4),("Z","YC","XC"))
def t(f,n=2000):
    f(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter()-t0)/n*1e6
for name,f in [("diff X",lambda: g.diff(T,"X")),("interp Y",lambda: g.interp(T,"Y")),("derivative X",lambda: g.derivative(T,"X")),("integrate Z",lambda: g.integrate(T,"Z")),("cumsum Z",lambda: g.cumsum(T,"Z")),("raw stencil1d",lambda: D.stencil1d("diff",T.data,2,1,0,"periodic"))]:
    print(name, round(t(f),1),"us/call")
import cProfile,pstats
pr=cProfile.Profile(); pr.enable()
for _ in range(1000): g.derivative(T,"X")
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(14)
