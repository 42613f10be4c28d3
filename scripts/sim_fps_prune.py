import sys, numpy as np
sys.path.insert(0,'/root/repo')
from ws3d_amd import synth
import oracle
def morton(cx, cz, bits):
    code = np.zeros_like(cx)
    for i in range(bits):
        code |= ((cx >> i) & 1) << (2*i) | ((cz >> i) & 1) << (2*i+1)
    return code
def run(kind, seed, grid=32, M=4096):
    xyz = synth.cloud(kind, 16384, seed)[:, :3].astype(np.float32)
    n=len(xyz)
    idx, temp = None, None
    seq = oracle.furthest_point_sample(xyz[None], M)[0]
    x,z = xyz[:,0],xyz[:,2]
    bits = int(np.log2(grid))
    cx = np.clip(((x-x.min())*grid/(x.max()-x.min())).astype(np.int64),0,grid-1); cz = np.clip(((z-z.min())*grid/(z.max()-z.min())).astype(np.int64),0,grid-1)
    order = np.argsort(morton(cx,cz,bits), kind='stable')
    P = xyz[order].astype(np.float64)
    def boxes(sz):
        nb = n//sz
        lo = P.reshape(nb,sz,3).min(1); hi = P.reshape(nb,sz,3).max(1)
        return lo,hi
    lo64,hi64 = boxes(64); lo16,hi16 = boxes(16)
    t = np.full(n,1e10)
    res = {}
    tot64=tot16slot=tot16=0; cnt=0; late64=late16slot=0
    perwave64 = np.zeros(16); 
    for j in range(1,M):
        q = xyz[seq[j-1]].astype(np.float64)
        d = ((P-q)**2).sum(1)
        # bucket max before update
        bm64 = t.reshape(-1,64).max(1); bm16 = t.reshape(-1,16).max(1)
        g64 = np.maximum(np.maximum(lo64-q, q-hi64),0); L64=(g64**2).sum(1)
        g16 = np.maximum(np.maximum(lo16-q, q-hi16),0); L16=(g16**2).sum(1)
        need64 = L64 < bm64
        need16 = L16 < bm16
        slot16 = need16.reshape(-1,4).any(1)
        t = np.minimum(t,d)
        if j>=64:
            tot64+=need64.sum(); tot16slot+=slot16.sum(); tot16+=need16.sum(); cnt+=1
    return tot64/cnt, tot16slot/cnt, tot16/cnt
for kind in ("hdl64","lidar"):
    for grid in (32,64,128):
        print(kind, "grid",grid, "needy 64-pt buckets/sample %.2f  slots with needy 16-pt sub-bucket %.2f  needy 16-pt buckets %.2f" % run(kind,3000,grid))
