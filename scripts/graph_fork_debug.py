"""Where does the fork/join capture of the Stage-1 step (side streams INSIDE the hipGraph) first differ from the single-stream one?
Records a clone of every SA / FP output, the heads and the proposals under four launch modes and compares them with the eager
single-stream pass: eager side streams, single-stream graph, fork/join graph (replayed several times, also with other work
in flight).  DESIGN.md 10.3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np, torch
from ws3d_amd import fastpath, synth, stage1, roipool3d_ops
from ws3d_amd.seeded import seeded_state_dict

torch.cuda.set_device(0)
cfg = stage1.DEFAULT_CFG
model = stage1.Stage1Net(mode="TEST").eval()
model.load_state_dict(seeded_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 7))
model = model.cuda()
pts = torch.from_numpy(np.stack([synth.cloud("hdl64", 16384, 3000 + s) for s in range(8)])).cuda()

LOG = []
orig_sa, orig_fp = fastpath.sa_forward, fastpath.fp_forward
def sa_tap(*a, **k):
    r = orig_sa(*a, **k); LOG.append(("sa_xyz", r[0].clone())); LOG.append(("sa_feat", r[1].clone())); return r
def fp_tap(*a, **k):
    r = orig_fp(*a, **k); LOG.append(("fp", r.clone())); return r
fastpath.sa_forward, fastpath.fp_forward = sa_tap, fp_tap

@torch.no_grad()
def body():
    LOG.clear()
    out = model.rpn_forward({"pts_input": pts})
    boxes, scores, count, enlarged = stage1.proposals_from_rpn(out, cfg, with_pool_boxes=True)
    feats = out["backbone_features"].transpose(1, 2).contiguous()
    pooled, empty = roipool3d_ops.roipool3d_gpu(out["backbone_xyz"], feats, boxes, cfg.roi_extra_width, sampled_pt_num=cfg.roi_sampled_pts, enlarged=enlarged)
    res = list(LOG) + [("rpn_cls", out["rpn_cls"]), ("rpn_reg", out["rpn_reg"]), ("boxes", boxes), ("scores", scores), ("count", count), ("pooled", pooled)]
    return res

def snapshot(res):
    torch.cuda.synchronize()
    return [(n, t.clone()) for n, t in res]

def compare(tag, ref, got):
    bad = [(i, n, float((a.float() - b.float()).abs().max())) for i, ((n, a), (_, b)) in enumerate(zip(ref, got)) if not torch.equal(a, b)]
    print("%-46s %s" % (tag, "identical to the eager single-stream pass (%d tensors)" % len(ref) if not bad else "FIRST DIFFERENCE at tensor %d (%s), max abs %.3g; %d of %d differ" % (bad[0][0], bad[0][1], bad[0][2], len(bad), len(ref))))
    return not bad

fastpath.GEOMETRY_AHEAD = False
for _ in range(2):
    body()
ref = snapshot(body())
fastpath.GEOMETRY_AHEAD = True
for _ in range(2):
    body()
ok_eager = all(compare("eager, side streams (pass %d)" % i, ref, snapshot(body())) for i in range(3))

def capture(fork):
    fastpath.GEOMETRY_AHEAD, fastpath.GEOMETRY_IN_CAPTURE = True, fork
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            body()
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        res = body()
    return g, res, s

def release_like_round_2(self):
    """round 2's _Geometry.release(): the side streams wait for an event of the caller's stream -- also inside a capture, i.e. AFTER
    they were joined: they re-enter the capture and nothing joins them again"""
    done = torch.cuda.Event()
    done.record(self.main)
    for st in self.side:
        st.wait_event(done)


release_now = fastpath._Geometry.release
for fork in (False, True, "with round 2's release()"):
    fastpath._Geometry.release = release_like_round_2 if isinstance(fork, str) else release_now
    try:
        g, res, s = capture(bool(fork))
    except Exception as e:
        print("capture(fork=%s) failed: %r" % (fork, e)); continue
    tag = ("graph, fork/join + re-fork after join" if isinstance(fork, str) else "graph, fork/join inside") if fork else "graph, single stream"
    for i in range(3):
        with torch.cuda.stream(s):
            g.replay()
        compare("%s (replay %d)" % (tag, i), ref, snapshot(res))
    # with a competing stream hammering the chip (widens any missing-dependency window)
    other = torch.cuda.Stream()
    x = torch.randn(4096, 4096, device="cuda")
    for i in range(3):
        with torch.cuda.stream(other):
            for _ in range(20):
                x = x @ x * 1e-3
        with torch.cuda.stream(s):
            g.replay()
        compare("%s (replay under load %d)" % (tag, i), ref, snapshot(res))
    try:
        import time
        torch.cuda.synchronize(); ts = []
        for _ in range(20):
            t0 = time.perf_counter()
            with torch.cuda.stream(s):
                g.replay()
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        print("%-46s replay latency median %.3f ms" % (tag, float(np.median(ts))))
    except Exception as e:
        print("timing failed", e)
