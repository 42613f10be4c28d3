import sys, numpy as np
sys.path.insert(0,'/root/repo')
from ws3d_amd import synth
def morton(cx, cz):
    code = np.zeros_like(cx)
    for i in range(5):
        code |= ((cx >> i) & 1) << (2*i) | ((cz >> i) & 1) << (2*i+1)
    return code
def sim(kind, seed, M=4096, NW=16, KMAX=8):
    xyz = synth.cloud(kind, 16384, seed)[:, :3].astype(np.float32)
    n = xyz.shape[0]
    x, z = xyz[:,0], xyz[:,2]
    cx = np.clip(((x-x.min())*32/(x.max()-x.min())).astype(np.int64),0,31); cz = np.clip(((z-z.min())*32/(z.max()-z.min())).astype(np.int64),0,31)
    order = np.argsort(morton(cx,cz), kind='stable')
    bucket_of = np.empty(n, np.int64); bucket_of[order] = np.arange(n)//64
    wave_of = bucket_of % NW
    t = np.full(n, 1e10, np.float32)
    def d2(q):
        d = xyz - xyz[q]
        return (d[:,2]*d[:,2] + (d[:,0]*d[:,0] + d[:,1]*d[:,1])).astype(np.float32)
    picks=[0]; t = np.minimum(t, d2(0))
    hist = np.zeros(KMAX+1, np.int64); rounds=0
    late_hist = np.zeros(KMAX+1, np.int64)
    while len(picks) < M:
        # per wave best and second best
        best_v = np.full(NW, -1.0, np.float32); best_i = np.zeros(NW, np.int64); sec_v = np.full(NW, -1.0, np.float32)
        for w in range(NW):
            idx = np.nonzero(wave_of==w)[0]
            tv = t[idx]
            a = np.argmax(tv); best_v[w]=tv[a]; best_i[w]=idx[a]
            tv2 = tv.copy(); tv2[a] = -1; sec_v[w] = tv2.max()
        accepted=[]; bound=-1.0; avail = np.ones(NW,bool)
        for k in range(KMAX):
            vv = np.where(avail, best_v, -2)
            cand = int(np.argmax(vv)); vk = vv[cand]
            if (vv==vk).sum()>1 and k>0: break
            if k>0:
                if not (vk > bound): break
                ok = True
                for a in accepted:
                    d = xyz[best_i[cand]] - xyz[best_i[a]]
                    dd = np.float32(d[2]*d[2] + (d[0]*d[0] + d[1]*d[1]))
                    if not (dd >= vk): ok=False; break
                if not ok: break
            accepted.append(cand); avail[cand]=False; bound=max(bound, sec_v[cand])
            if len(picks)+len(accepted) >= M: break
        for a in accepted:
            picks.append(int(best_i[a])); t = np.minimum(t, d2(best_i[a]))
        hist[len(accepted)] += 1; rounds += 1
        if len(picks) > 1024: late_hist[len(accepted)] += 1
    return picks, hist, rounds, late_hist
for kind in ("hdl64","lidar"):
    for KMAX in (4,8):
        p,h,r,lh = sim(kind, 3000, KMAX=KMAX)
        print(kind, "KMAX",KMAX,"rounds",r,"picks/round %.2f"%(4095/r),"hist",h.tolist())
# verify equals plain FPS
def plain(kind, seed, M=4096):
    xyz = synth.cloud(kind, 16384, seed)[:, :3].astype(np.float32)
    t=np.full(16384,1e10,np.float32); picks=[0]
    for j in range(1,M):
        d=xyz-xyz[picks[-1]]; dd=(d[:,2]*d[:,2]+(d[:,0]*d[:,0]+d[:,1]*d[:,1])).astype(np.float32); t=np.minimum(t,dd); picks.append(int(np.argmax(t)))
    return picks
p,h,r,lh = sim("hdl64",3000,KMAX=8); print("same as plain FPS:", p==plain("hdl64",3000))
