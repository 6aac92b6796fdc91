"""The shape envelope of StackGRUDecoder(params): the reference builds its decoder from an arbitrary
model_params.json (examples/IC50/train_paccmann_rl.py:93-97), so hidden sizes between the
power-of-two ones and SELFIES alphabets of a few hundred tokens have to work -- to the same bar as
the BASELINE configurations: greedy tokens exact, sampled tokens = the oracle's inverse-CDF choice for
the same Philox uniform, log-probs and every gradient within 1e-4 of the fp64 oracle."""
import pytest
import torch

from tests import parity as P

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures('forward_path')]
TOL = 1e-4
MID = dict(embedding_size=32, stack_width=20, stack_depth=12, n_layers=1, latent_dim=64, pad_index=0, vocab_size=40)


@pytest.mark.parametrize('H', [384, 640, 768, 896])
def test_hidden_sizes_between_the_powers_of_two(gpu, H):
    cfg = dict(MID, rnn_cell_size=H)
    P.check_train(cfg, B=19, T=11, tol=TOL)
    assert P.check_greedy(cfg, 21, 14)['mismatched_rows'] == 0
    P.check_step(cfg, 18)


@pytest.mark.parametrize('H', [384, 768])
def test_hidden_384_768_two_layers_and_reference_stack(gpu, H):
    cfg = dict(P.CFG1, rnn_cell_size=H, n_layers=2)      # E=128, stack 50x50, V=60 as the BASELINE configs
    P.check_train(cfg, B=18, T=30, tol=TOL)
    r = P.check_greedy(cfg, 20, 40)
    assert r['mismatched_rows'] == r['near_ties']


@pytest.mark.parametrize('V', [300, 575])
@pytest.mark.parametrize('H', [128, 512])
def test_vocabularies_of_a_few_hundred_tokens(gpu, V, H):
    cfg = dict(MID, rnn_cell_size=H, vocab_size=V)
    P.check_train(cfg, B=21, T=9, tol=TOL)
    assert P.check_greedy(cfg, 19, 12)['mismatched_rows'] == 0
    r = P.check_sampling(cfg, 18, 10, temperature=1.2)      # V > 256: the two-pass CDF of the sampler
    assert r['exact'] + r['boundary'] == 18 * 10 and r['boundary'] <= 2
    P.check_step(cfg, 17)


@pytest.mark.parametrize('H,V,W,D,B,T', [
    (256, 180, 20, 12, 33, 40),     # 3+W+V in (128, 256]: the 1-KB dx records of the clustered BPTT; a cluster with one row
    (128, 40, 64, 2, 70, 25),       # widest stack, shallowest: 8 stack rows per pass of the 512 threads, 2 sequences per owner
    (512, 60, 50, 50, 128, 70),     # the largest clustered launch at hidden 512 (8 clusters of 32 CUs), T past the depth
    (384, 97, 33, 7, 96, 19),       # 24 CUs per cluster, odd sizes everywhere
])
def test_small_batch_cluster_edges(gpu, H, V, W, D, B, T):
    cfg = dict(MID, rnn_cell_size=H, vocab_size=V, stack_width=W, stack_depth=D)
    P.check_train(cfg, B=B, T=T, tol=TOL)
    assert P.check_greedy(cfg, B, T)['mismatched_rows'] == 0


@pytest.mark.parametrize('H,B', [(256, 300), (256, 1000), (512, 200), (512, 700), (384, 250), (384, 890)])
def test_several_row_groups_per_cluster(gpu, H, B):
    """More row groups than resident clusters: a cluster serves several (pmg_fwdc.hip / pmg_bwdc.hip), a member
    owns several sequences; where the BPTT's owned sequences no longer fit the LDS (B=1000 at hidden 256, 890 at
    384) the clustered forward is paired with the batch-partitioned BPTT."""
    cfg = dict(MID, rnn_cell_size=H)
    P.check_train(cfg, B=B, T=14, tol=TOL)
    assert P.check_greedy(cfg, B, 16)['mismatched_rows'] == 0
    r = P.check_sampling(cfg, B, 12, temperature=1.1)
    assert r['exact'] + r['boundary'] == B * 12 and r['boundary'] <= 2


def test_reference_stack_hidden_512_three_owned_sequences(gpu):
    cfg = dict(P.CFG1, rnn_cell_size=512)       # stack 50x50: six row groups per cluster is what the LDS holds
    P.check_train(cfg, B=760, T=60, tol=TOL)


def test_history_layout_follows_the_forward(gpu, forward_path):
    """The clustered forward leaves stack / push / probability history per sequence when the clustered BPTT
    will read it (FwdArgs::LAYW).  A BPTT told PMG_FLAG_NO_CLUSTER on such a workspace must not return
    plausible numbers; a clustered BPTT after a batch-partitioned forward reads the interleaved layout."""
    import ctypes as C
    from paccmann_generator_amd import _ffi, ops
    if forward_path != 'clustered':
        pytest.skip('one pass is enough')
    cfg = dict(MID, rnn_cell_size=256)
    _, pdec = P.make_pair(cfg, seed=2)
    w = pdec.weights()
    B, T = 40, 12
    tokens = P.synthetic_tokens(T, B, cfg['vocab_size'], seed=3).cuda()
    torch.manual_seed(4)
    h0 = torch.tanh(torch.randn(1, B, 256)).cuda()
    reward = torch.rand(B).cuda()
    params = [p.detach() for p in w.flat()]
    d = ops._dims_of(params, w.dims['D'], w.pad_idx)
    shape = ops._shape(d, B, T)
    wstruct, keep = ops._wstruct(params, 1)

    def run(fwd_flags, bwd_flags):
        ws = ops._alloc_ws(shape, _ffi.PMG_MODE_TRAIN, h0.device)
        logp = torch.empty(T - 1, B, device='cuda')
        _ffi.check(_ffi.lib.pmg_logprob_fwd(C.byref(shape), C.byref(wstruct), _ffi._ptr(tokens), _ffi._ptr(h0),
                                            _ffi._ptr(logp), None, fwd_flags, _ffi._ptr(ws), _ffi.current_stream()), 'fwd')
        grads = [torch.empty_like(p) for p in params]
        gstruct, keep2 = ops._grad_struct(grads, 1)
        gh0 = torch.empty_like(h0)
        loss = torch.empty(1, device='cuda')
        _ffi.check(_ffi.lib.pmg_reinforce_bwd(C.byref(shape), C.byref(wstruct), _ffi._ptr(tokens), _ffi._ptr(reward),
                                              1.0 / B, bwd_flags, C.byref(gstruct), _ffi._ptr(gh0), _ffi._ptr(loss),
                                              _ffi._ptr(ws), _ffi.current_stream()), 'bwd')
        torch.cuda.synchronize()
        return loss.item(), gh0, grads

    NC = _ffi.PMG_FLAG_NO_CLUSTER
    ref = run(NC, NC)
    for ff, bf in ((0, 0), (NC, 0)):
        l, g, gs = run(ff, bf)
        assert abs(l - ref[0]) <= 1e-5 * abs(ref[0])
        assert torch.allclose(g, ref[1], rtol=1e-4, atol=1e-7)
        assert all(torch.allclose(x, y, rtol=2e-4, atol=1e-6) for x, y in zip(gs, ref[2]))
    l, g, _ = run(0, NC)
    assert l != l and bool(torch.isnan(g).all())


def test_wide_vocabulary_with_the_reference_stack_and_hidden_768(gpu):
    cfg = dict(P.CFG1, rnn_cell_size=768, vocab_size=575)
    P.check_train(cfg, B=17, T=20, tol=TOL)
    assert P.check_greedy(cfg, 18, 24)['mismatched_rows'] == 0


def test_out_of_range_token_ids_are_clamped_consistently(gpu):
    """ids outside [0, V) are rejected on the host by ops.reinforce_loss; a caller that skips that check
    (trusted_tokens=True) gets them clamped the same way by the forward, the BPTT and the
    weight-gradient kernels, i.e. exactly the result of the clamped ids."""
    from paccmann_generator_amd import _ffi, ops
    _, pdec = P.make_pair(dict(P.TINY, vocab_size=97), seed=0)       # V > 64: the one-hot job reads the ids too
    tokens = P.synthetic_tokens(8, 20, 97, seed=4).cuda()
    bad = tokens.clone()
    bad[2, 3], bad[4, 7], bad[5, 0] = 97, 4000, -5
    torch.manual_seed(1)
    h0 = torch.tanh(torch.randn(1, 20, 128)).cuda()
    reward = torch.rand(20).cuda()
    with pytest.raises(_ffi.PmgError):
        ops.reinforce_loss(pdec.weights(), bad, h0, reward, 0.05)
    outs = []
    for t in (bad, bad.clamp(0, 96)):
        for p in pdec.parameters():
            p.grad = None
        h = h0.clone().requires_grad_(True)
        loss, logp = ops.reinforce_loss(pdec.weights(), t, h, reward, 0.05, trusted_tokens=True)
        loss.backward()
        outs.append((loss.item(), logp.clone(), h.grad.clone(), [p.grad.clone() for p in pdec.parameters() if p.grad is not None]))
    (l0, lp0, g0, ps0), (l1, lp1, g1, ps1) = outs
    assert l0 == l1 and torch.equal(lp0, lp1) and torch.equal(g0, g1)
    assert all(torch.equal(a, b) for a, b in zip(ps0, ps1))
