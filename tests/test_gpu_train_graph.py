"""graphs.TrainStepGraph: the training step replayed from hipGraphs must be the eager step -- same losses, same
parameters after the same number of steps -- with and without a GradientExchange, on new input batches, and must not leave
stale packed weights behind for the next eager forward."""
import copy

import pytest
import torch

from wavelet_monodepth_amd import synth
from util import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _network(dev, layers=18, seed=1):
    from wavelet_monodepth_amd.encoders import ResnetEncoder
    from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder
    torch.manual_seed(seed)
    enc = ResnetEncoder(layers).to(dev)
    dec = synth.fill_state_dict(DepthWaveProgressiveDecoder(enc.num_ch_enc), seed=seed).to(dev)
    return enc, dec


def _batch(dev, tag, B=2, H=96, W=160):
    img = torch.from_numpy(synth.uniform((B, 3, H, W), "tg_img" + tag, 0, 0.0, 1.0)).to(dev)
    tgt = [torch.from_numpy(synth.uniform((B, 1, H >> s, W >> s), "tg_tgt" + tag, s, 0.05, 0.95)).to(dev) for s in range(4)]
    return [img] + tgt


def _loss_fn(enc, dec):
    def fn(img, *tgt):
        out = dec(enc(img))
        return sum((out[("disp", s)] - tgt[s]).abs().mean() for s in range(4))
    return fn


def _optimizer(kind, params):
    if kind == "sgd":
        return torch.optim.SGD(params, lr=1e-3, momentum=0.9)
    return torch.optim.Adam(params, lr=1e-4, capturable=True)


def _same(got, want, what):
    """2e-3 of the tensor's magnitude plus 2e-7: parameters that start at zero (BatchNorm biases) hold nothing but
    lr * sum of gradients after a few steps, and MIOpen's reductions (atomics) round differently from run to run."""
    err = float((got.detach() - want.detach()).abs().max())
    bound = 2e-3 * float(want.detach().abs().max()) + 2e-7
    assert err <= bound, "%s: max abs error %.3e > %.3e" % (what, err, bound)


def _params(*mods):
    return [p for m in mods for p in m.parameters()]


@pytest.mark.parametrize("kind", ["sgd", "adam"])
def test_replayed_steps_equal_eager_steps(dev, kind):
    from wavelet_monodepth_amd.graphs import TrainStepGraph
    enc_a, dec_a = _network(dev)
    enc_b, dec_b = copy.deepcopy(enc_a), copy.deepcopy(dec_a)
    batches = [_batch(dev, str(i)) for i in range(3)]
    # a: five eager steps (two on batch 0 = the warm-up of b, then batches 0, 1, 2)
    opt_a = _optimizer(kind, _params(enc_a, dec_a))
    fn_a = _loss_fn(enc_a, dec_a)
    losses_a = []
    for b in [batches[0], batches[0]] + batches:
        opt_a.zero_grad(set_to_none=True)
        loss = fn_a(*b)
        loss.backward()
        opt_a.step()
        losses_a.append(float(loss))
    # b: two warm-up steps inside the constructor, three replays with fresh inputs copied into the static buffers
    opt_b = _optimizer(kind, _params(enc_b, dec_b))
    g = TrainStepGraph(_loss_fn(enc_b, dec_b), opt_b, inputs=[t.clone() for t in batches[0]], warmup=2)
    losses_b = [float(g.step(*b)) for b in batches]
    tol = 1e-4 if kind == "sgd" else 2e-3      # Adam divides by sqrt(v): rounding noise of near-zero gradients is amplified
    for la, lb in zip(losses_a[2:], losses_b):
        assert abs(la - lb) <= tol * abs(la), (losses_a, losses_b)
    assert losses_b[0] != losses_b[1]          # the replay really read the new batch
    if kind == "sgd":
        for (n, pa), pb in zip(list(enc_a.named_parameters()) + list(dec_a.named_parameters()), _params(enc_b, dec_b)):
            _same(pb, pa, n)
        for (n, ba), bb in zip(enc_a.named_buffers(), enc_b.buffers()):      # BatchNorm statistics are updated inside the graph
            if ba.dtype == torch.float32:
                _same(bb, ba, n)
            else:
                assert int(ba) == int(bb) == 5, n


def test_replay_with_gradient_exchange_and_no_stale_weight_images(dev):
    from wavelet_monodepth_amd import ops
    from wavelet_monodepth_amd.ddp import GradientExchange, bucket_groups
    from wavelet_monodepth_amd.graphs import TrainStepGraph
    enc_a, dec_a = _network(dev, seed=2)
    enc_b, dec_b = copy.deepcopy(enc_a), copy.deepcopy(dec_a)
    batch = _batch(dev, "x")
    opt_a = _optimizer("sgd", _params(enc_a, dec_a))
    fn_a = _loss_fn(enc_a, dec_a)
    for _ in range(4):
        opt_a.zero_grad(set_to_none=True)
        fn_a(*batch).backward()
        opt_a.step()
    gx = GradientExchange(bucket_groups(enc_b, dec_b, bucket_bytes=4 << 20), world=1, rank=0, backend="torch")
    assert len(gx.buckets) > 3
    opt_b = _optimizer("sgd", _params(enc_b, dec_b))
    g = TrainStepGraph(_loss_fn(enc_b, dec_b), opt_b, inputs=batch, exchange=gx, warmup=1)
    feats_eval = [f.detach() for f in enc_b.eval()(batch[0])]
    with torch.no_grad():
        dec_b(feats_eval)                       # memoises packed weights of the current parameters
    enc_b.train()
    for _ in range(3):
        g.step()
    for (n, pa), pb in zip(list(enc_a.named_parameters()) + list(dec_a.named_parameters()), _params(enc_b, dec_b)):
        _same(pb, pa, n)
    # the eager forward after the replays must use the replayed weights, not the memoised images of the old ones
    with torch.no_grad():
        got = dec_b(feats_eval)
        ops.invalidate_packs()
        want = dec_b(feats_eval)
        ref = dec_a(feats_eval)
    for k in want:
        if isinstance(k, tuple) and k[0] == "disp":
            assert torch.equal(got[k], want[k]), k
            assert_close(got[k], ref[k], 1e-3, str(k))
    gx.close()


_STALE = r"""
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import test_gpu_train_graph as T
from wavelet_monodepth_amd.graphs import TrainStepGraph
dev = torch.device("cuda:0")
enc, dec = T._network(dev)
batch = T._batch(dev, "s")
fn = T._loss_fn(enc, dec)
opt = T._optimizer("sgd", T._params(enc, dec))
loss = fn(*batch); loss.backward(); opt.step()          # eager step on the default stream; `loss` and dec.outputs stay alive
try:
    TrainStepGraph(fn, opt, inputs=batch, warmup=1)
except RuntimeError as e:
    print("REFUSED:", str(e)[:80])
    loss = None
    g = TrainStepGraph(fn, opt, inputs=batch, warmup=1, modules=[enc, dec])
    print("CAPTURED", float(g.step()))
"""


def test_a_live_eager_graph_is_reported_not_captured(dev):
    """Eager steps on the default stream leave AccumulateGrad nodes bound to it for as long as their graph lives; capturing
    through them kills the process on ROCm.  Runs in a child process for that reason."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([sys.executable, "-W", "ignore", "-c", _STALE % (os.path.dirname(here), here)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-600:]
    assert "REFUSED: TrainStepGraph: an autograd graph of an earlier eager step" in p.stdout, p.stdout
    assert "CAPTURED" in p.stdout
