"""Stage-1 training targets / losses against the fixture produced by the reference's
own label generator and model_fn (tests/golden/make_golden_train.py -> train_losses.json), plus the
trainer's schedules and checkpoint format.  CPU only."""
import json
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

from ws3d_amd import losses  # noqa: E402


@pytest.fixture(scope="module")
def fx():
    with open(os.path.join(HERE, "golden", "train_losses.json")) as f:
        return json.load(f)


def _inputs(case):
    # same seeded inputs as the generator (its helper only needs numpy + ws3d_amd.synth)
    from ws3d_amd import synth
    B, n = case["batch"], case["n"]
    pc = synth.make_batch("lidar", B, n, case["config_id"])
    centres = [synth.random_boxes3d(15, (1000 * case["config_id"] + b) * 7919 + 13)[:case["cars"], :3].astype(np.float32)
               for b in range(B)]
    rng = np.random.default_rng(case["config_id"])
    return pc, centres, rng.normal(-2.0, 1.5, (B, n, 1)).astype(np.float32), rng.normal(0.0, 1.0, (B, n, 40)).astype(np.float32)


@pytest.mark.parametrize("name", ["two_scenes", "no_centres"])
def test_labels_and_loss_match_reference(fx, name):
    ref = fx["cases"][name]
    pc, centres, rpn_cls, rpn_reg = _inputs(ref["case"])
    B = pc.shape[0]
    labels = [losses.gaussian_center_labels(pc[b, :, :3], centres[b]) for b in range(B)]
    cls_label = np.stack([np.asarray(l[0], dtype=np.float64) for l in labels])
    reg_label = np.stack([l[1] for l in labels])
    np.testing.assert_allclose(cls_label.reshape(-1)[np.array(ref["cls_label"]["pos"])], ref["cls_label"]["val"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(cls_label.sum(), ref["cls_label"]["sum"], rtol=1e-12)
    assert int((cls_label > 0).sum()) == ref["cls_label"]["fg"]
    import hashlib
    assert hashlib.sha256(np.ascontiguousarray(reg_label).tobytes()).hexdigest() == ref["reg_label"]["sha256"]
    assert int((reg_label != 0).sum()) == ref["reg_label"]["nonzero"]

    t_cls = torch.from_numpy(rpn_cls).requires_grad_(True)
    t_reg = torch.from_numpy(rpn_reg).requires_grad_(True)
    loss, tb = losses.rpn_loss(t_cls, t_reg, torch.from_numpy(cls_label).float(), torch.from_numpy(reg_label).float(),
                               loc_scope=4.0, loc_bin_size=0.8)
    loss.backward()
    assert loss.item() == pytest.approx(ref["loss"], rel=1e-6)
    for k, v in ref["tb"].items():
        assert tb[k] == pytest.approx(v, rel=2e-6, abs=1e-7), k
    g = t_cls.grad.numpy().reshape(-1)
    np.testing.assert_allclose(g[np.array(ref["grad_cls"]["pos"])], ref["grad_cls"]["val"], rtol=1e-5, atol=1e-9)
    gr = (t_reg.grad.numpy() if t_reg.grad is not None else np.zeros_like(rpn_reg)).reshape(-1)
    np.testing.assert_allclose(gr[np.array(ref["grad_reg"]["pos"])], ref["grad_reg"]["val"], rtol=1e-5, atol=1e-9)


def test_focal_and_reg_loss_properties():
    logits = torch.tensor([-3.0, 0.0, 2.5, 8.0])
    w = torch.ones(4)
    hard = losses.sigmoid_focal_loss(logits, torch.tensor([0.0, 1.0, 1.0, 1.0]), w)
    assert (hard >= 0).all() and hard[3] < hard[2] < hard[1]          # confident positives are down-weighted
    assert losses.sigmoid_focal_loss(logits, torch.tensor([0.0, 1.0, 1.0, 1.0]), w, gamma=0, alpha=None)[1].item() == \
        pytest.approx(np.log(2.0), rel=1e-6)
    # a prediction that puts all mass on the right bins with the right residual has ~zero loss
    loc_scope, bin_size = 4.0, 0.8
    lab = torch.tensor([[1.3, 0.0, -2.2], [-3.9, 0.0, 3.99]])
    bins = 10
    pred = torch.zeros((2, 40))
    for i in range(2):
        for col, lo in ((0, 0), (2, bins)):
            shift = min(max(lab[i, col].item() + loc_scope, 0), 2 * loc_scope - 1e-3)
            b = int(shift // bin_size)
            pred[i, lo + b] = 50.0
            pred[i, 2 * bins + lo + b] = (shift - (b * bin_size + bin_size / 2)) / (bin_size / 2)
    total, parts = losses.rpn_reg_loss(pred, lab, loc_scope, bin_size)
    assert total.item() < 1e-6 and set(parts) == {"loss_x_bin", "loss_z_bin", "loss_x_res", "loss_z_res"}


# ----------------------------------------------------------------------------- trainer pieces (CPU)
def test_one_cycle_schedule_matches_reference_formula():
    """against an independent restatement of learning_schedules_fastai.OneCycle (phases from
    int(pct*total), cosine annealing, floor 2e-6)"""
    from ws3d_amd.train_rpn import one_cycle
    total, lr_max, moms, div, pct = 200, 0.002, (0.95, 0.85), 10.0, 0.4
    a1 = int(total * pct)
    cosa = lambda s, e, p: e + (s - e) / 2 * (np.cos(np.pi * p) + 1)
    for step in range(total):
        lr = cosa(lr_max / div, lr_max, step / a1)
        mom = cosa(moms[0], moms[1], step / a1)
        if step >= a1:
            lr, mom = cosa(lr_max, 2e-6, (step - a1) / (total - a1)), cosa(moms[1], moms[0], (step - a1) / (total - a1))
        got = one_cycle(step, total, lr_max, moms, div, pct)
        assert got[0] == pytest.approx(lr, rel=1e-12) and got[1] == pytest.approx(mom, rel=1e-12)
    assert one_cycle(0, total, lr_max)[0] == pytest.approx(lr_max / 10) and one_cycle(a1, total, lr_max)[0] == pytest.approx(lr_max)
    assert one_cycle(total - 1, total, lr_max)[0] < 1e-5


def test_bn_momentum_decay_and_decoupled_weight_decay():
    from ws3d_amd.train_rpn import AdamOneCycle, TrainConfig, bn_momentum_at, set_bn_momentum
    cfg = TrainConfig()
    assert bn_momentum_at(0, cfg) == 0.1 and bn_momentum_at(999, cfg) == 0.1 and bn_momentum_at(1000, cfg) == 0.05
    assert bn_momentum_at(10 ** 9, TrainConfig(bn_decay_step_list=tuple(range(1, 30)))) == cfg.bnm_clip
    net = torch.nn.Sequential(torch.nn.Conv1d(3, 4, 1), torch.nn.BatchNorm1d(4))
    set_bn_momentum(net, 0.03)
    assert net[1].momentum == 0.03
    # one step with zero gradient: Adam's update is 0, the decoupled decay alone shrinks the weights
    p = torch.nn.Parameter(torch.ones(5))
    opt = AdamOneCycle([p], total_step=100, cfg=cfg)
    opt.schedule(0)
    p.grad = torch.zeros(5)
    opt.step()
    np.testing.assert_allclose(p.detach().numpy(), 1 - cfg.weight_decay * (cfg.lr / cfg.div_factor), rtol=1e-6)
    assert opt.opt.param_groups[0]["betas"] == (cfg.moms[0], 0.99)


def test_checkpoint_format_and_reference_keys(tmp_path):
    """{'it','model_state','optimizer_state'} in '<name>.pth', model_state keyed like the reference's
    PointRCNN (fixture tests/golden/stage1_state_dict.json), loadable without the optimizer"""
    from ws3d_amd import stage1
    from ws3d_amd.train_rpn import AdamOneCycle, checkpoint_state, load_checkpoint, save_checkpoint
    small = stage1.RPNConfig(npoints=(64, 32, 16, 8), num_points=256)
    model = stage1.Stage1Net(mode="TRAIN", cfg=small)
    opt = AdamOneCycle(model.parameters(), 10)
    path = save_checkpoint(checkpoint_state(model, opt, 7), str(tmp_path / "checkpoint_iter_00007"))
    assert path.endswith("checkpoint_iter_00007.pth")
    ck = torch.load(path, map_location="cpu")
    assert set(ck) == {"it", "model_state", "optimizer_state"} and ck["it"] == 7
    ref_keys = json.load(open(os.path.join(HERE, "golden", "stage1_state_dict.json")))
    ref_keys = ref_keys["keys"] if isinstance(ref_keys, dict) and "keys" in ref_keys else ref_keys
    assert list(ck["model_state"].keys()) == list(ref_keys)
    other = stage1.Stage1Net(mode="TEST", cfg=small)
    it, epoch = load_checkpoint(other, None, path)
    assert (it, epoch) == (7, -1)
    for k, v in model.state_dict().items():
        assert torch.equal(v, other.state_dict()[k])
    with pytest.raises(FileNotFoundError):
        load_checkpoint(other, None, str(tmp_path / "missing.pth"))
    # the same optimizer layout resumes; a reference-style optimizer_state (fastai OptimWrapper: two parameter groups) is
    # skipped with a warning instead of raising, model_state and `it` are still restored
    opt2 = AdamOneCycle(other.train().parameters(), 10)
    assert load_checkpoint(other, opt2, path)[0] == 7
    n = len(ck["optimizer_state"]["param_groups"][0]["params"])
    ck["optimizer_state"]["param_groups"] = [dict(ck["optimizer_state"]["param_groups"][0], params=list(range(n // 2))),
                                             dict(ck["optimizer_state"]["param_groups"][0], params=list(range(n // 2, n)))]
    torch.save(ck, str(tmp_path / "ref_style.pth"))
    with pytest.warns(UserWarning, match="parameter-group layout"):
        assert load_checkpoint(other, AdamOneCycle(other.parameters(), 10), str(tmp_path / "ref_style.pth"))[0] == 7


def test_scene_augmentation_matches_reference_stream(fx):
    """same numpy seed -> same rotation angle, scale and flip decision, bit-identical points and boxes"""
    import hashlib
    from ws3d_amd import synth
    for ref in fx["augmentation"]:
        pts = synth.lidar_cloud(512, 900 + ref["seed"])[:, :3].astype(np.float64)
        boxes = synth.random_boxes3d(6, 77 + ref["seed"]).astype(np.float64)
        np.random.seed(ref["seed"])
        a_pts, a_box, methods = losses.scene_augmentation(pts, boxes)
        assert [m if isinstance(m, str) else [m[0], float(m[1])] for m in methods] == ref["methods"]
        assert hashlib.sha256(np.ascontiguousarray(a_pts).tobytes()).hexdigest() == ref["pts_sha256"]
        assert hashlib.sha256(np.ascontiguousarray(a_box).tobytes()).hexdigest() == ref["box_sha256"]
    # inputs are not modified; centre-only annotations (K,3) are accepted
    pts = np.ones((4, 3)); c = np.ones((2, 3))
    losses.scene_augmentation(pts, c, np.random.RandomState(0))
    assert (pts == 1).all() and (c == 1).all()


def test_loader_processes_deliver_the_same_batches():
    """batches(workers=2): forked loader processes prepare the scenes; composition and order of the
    mini-batches equal the in-process loader's (a dataset without random draws gives equal data)"""
    from ws3d_amd.train_rpn import SyntheticCenters, batches
    ds = SyntheticCenters(10, npoints=512)
    a = batches(ds, 3, np.random.RandomState(4))
    b = batches(ds, 3, np.random.RandomState(4), workers=2, ahead=2)
    try:
        for _ in range(7):                      # crosses an epoch boundary (3 batches per epoch, drop_last)
            x, y = next(a), next(b)
            assert x["sample_id"] == y["sample_id"]
            for k in ("pts_input", "rpn_cls_label", "rpn_reg_label"):
                np.testing.assert_array_equal(x[k], y[k])
    finally:
        a.close(); b.close()
    # two ranks take disjoint slices of each global batch, also with loader processes
    r0 = batches(ds, 2, np.random.RandomState(1), rank=0, world=2, workers=1)
    r1 = batches(ds, 2, np.random.RandomState(1), rank=1, world=2, workers=1)
    try:
        for _ in range(3):
            assert not set(next(r0)["sample_id"]) & set(next(r1)["sample_id"])
    finally:
        r0.close(); r1.close()


def test_item_batches_finite_source_and_worker_errors():
    """ws3d_amd.loader: a finite list of index batches is served completely and in order by the loader
    processes; an exception raised inside a loader process surfaces in the consumer"""
    from ws3d_amd import loader

    class Squares:
        def __getitem__(self, i):
            if i == 13:
                raise ValueError("bad scene 13")
            return {"i": i, "sq": np.full((4,), i * i)}

    ids = [[0, 1, 2], [3], [4, 5], [], [6]]
    for workers in (0, 2):
        got = list(loader.item_batches(Squares(), ids, workers=workers, ahead=2, seed=1))
        assert [[it["i"] for it in b] for b in got] == ids
        assert all(int(it["sq"][0]) == it["i"] ** 2 for b in got for it in b)
        with pytest.raises(ValueError, match="bad scene 13"):
            list(loader.item_batches(Squares(), [[1], [13], [2]], workers=workers, ahead=2, seed=1))
