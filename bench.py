#!/usr/bin/env python
"""Headline benchmark: sampled SMILES tokens/s for one decode + REINFORCE step.

A "step" = one pass of the hot path over one batch of synthetic latents
(BASELINE.json configs[1]: hidden 512, stack depth 50, batch 4096 per GPU, len 128, fp32):
  1. sampling decode of B x T tokens                  (pmg_decode)
  2. append 2 stop tokens, pad (device twin of reinforce.py:177-184 / pad_sequence)
  3. h0 = latent_to_hidden(latent)                    (torch, autograd-connected)
  4. teacher-forced log-probs                         (pmg_logprob_fwd)
  5. reward-weighted BPTT + weight gradients          (pmg_reinforce_bwd)
  6. N>1: one RCCL all-reduce of the flat gradient
  7. clip_grad_norm_ + Adam                           (torch)
Rewards are synthetic and resident (RDKit scoring stays on the host and is not part of
the step).  Random-init weights, synthetic latents.

Launch: python bench.py [--gpus N --steps K --warmup W].  With N>1 and no WORLD_SIZE in the
environment it starts its own N ranks (python -m torch.distributed.run --nproc-per-node N ...);
launched under torch.distributed.run it is one of the ranks.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md, "Peak FP32 (matrix)"

CONFIGS = {
    'cfg1': dict(embedding_size=128, rnn_cell_size=256, vocab_size=60, stack_width=50, stack_depth=50,
                 n_layers=1, latent_dim=256, pad_index=0, batch=64, length=100),
    'cfg2': dict(embedding_size=128, rnn_cell_size=512, vocab_size=60, stack_width=50, stack_depth=50,
                 n_layers=1, latent_dim=256, pad_index=0, batch=4096, length=128),
    # configs[3] at its stated size; the training pair runs as 4 chunks of 4096 sequences whose gradients
    # accumulate (ops.plan_chunks: 80 GB of history per chunk)
    'cfg4': dict(embedding_size=128, rnn_cell_size=1024, vocab_size=60, stack_width=50, stack_depth=50,
                 n_layers=2, latent_dim=256, pad_index=0, batch=16384, length=256),
    # configs[4]: teacher-forced log-prob + BPTT only, over pre-sampled sequences in chunks of `batch`
    'cfg5': dict(embedding_size=128, rnn_cell_size=512, vocab_size=60, stack_width=50, stack_depth=50,
                 n_layers=1, latent_dim=256, pad_index=0, batch=4096, length=128),
}
KIND = {'cfg1': 'omics-conditioned (reinforce_omics path)', 'cfg2': 'omics-conditioned',
        'cfg4': 'protein-conditioned (reinforce_proteins path)',
        'cfg5': '1M-class pre-sampled sequences, teacher-forced log-prob + BPTT only,'}
TAIL = {True: '', False: ' decode+REINFORCE,'}
START, STOP = 2, 3


def flops_per_token(c):
    """forward flops per (sequence, step)."""
    H, E, W, D, V, L = (c['rnn_cell_size'], c['embedding_size'], c['stack_width'], c['stack_depth'],
                        c['vocab_size'], c['n_layers'])
    f = 0
    for l in range(L):
        i_l = (E + W) if l == 0 else H
        f += 6 * H * (i_l + H)
    return f + 6 * H + 2 * H * W + 5 * D * W + 2 * H * V


class _Lang:
    start_index = START
    stop_index = STOP


class _Log:
    def info(self, *a, **k):
        pass

    warning = info


def build_learner(cfg, device, process_group=None):
    from paccmann_generator_amd import Generator, StackGRUDecoder
    from paccmann_generator_amd.reinforce import Reinforce
    torch.manual_seed(0)
    dec = StackGRUDecoder(cfg)
    dec = dec.to(device).eval()
    gen = Generator(None, dec)
    gen._associate_language(_Lang())
    gen.seed(1234)
    params = dict(generate_len=cfg['length'], temperature=0.8, clip_grad=0.1, learning_rate=3e-4,
                  save_untrained=False)
    learner = Reinforce(gen, None, params, 'bench', _Log(), is_valid=lambda s: True,
                        reward_fn=None, process_group=process_group)
    return learner


def cpu_baseline(cfg, budget_s=12.0):
    """The CPU oracle driven through the reference's loop structure (generate +
    policy_gradient), on a bounded sample of the same workload."""
    from oracle import reinforce_loop as RL
    from oracle.stack_gru import OracleGenerator, SamplingSearch, build_decoder
    B = 64 if cfg['rnn_cell_size'] <= 512 else 16
    T = cfg['length']
    dec = build_decoder(cfg, seed=0)
    gen = OracleGenerator(dec)
    opt = RL.make_optimizer(dec, dict(learning_rate=3e-4))
    torch.manual_seed(1)
    latent = torch.randn(1, B, cfg['latent_dim'])
    torch.manual_seed(2)
    rewards = torch.rand(B).numpy()

    def one():
        mols = list(gen.generate(latent, torch.tensor([START]), torch.tensor([STOP]), T, SamplingSearch(0.8)))
        nums = RL.append_stops(mols, STOP)
        RL.policy_gradient_step(dec, opt, latent, nums, list(range(B)), rewards, lrps=1, clip_grad=0.1)

    one()  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        one()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 60:
            break
    return dict(value=B * T * n / el, unit='tokens/s', cores=torch.get_num_threads(), kind='port',
                sample=f'{n} decode+REINFORCE steps of the eager-PyTorch oracle at B={B}, T={T} '
                       f'(same model shape), {el:.1f}s')


def presampled_tokens(n, T, V, device, seed=5):
    """uniform random tokens in [4,V), length ~U[20, T-2], then stop, stop, pad.
    Returns [T+2, n] int64 (pad_sequence layout)."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    lens = torch.randint(20, T - 1, (n,), generator=g)
    body = torch.randint(4, V, (T + 2, n), generator=g)
    pos = torch.arange(T + 2).unsqueeze(1)
    toks = torch.where(pos < lens, body, torch.where(pos < lens + 2, torch.full_like(body, STOP), torch.zeros_like(body)))
    return toks.to(device)


def kernel_source_hash():
    """sha256 over the HIP sources and headers the library is built from."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, 'paccmann_generator_amd', 'csrc')
    for path in sorted(os.path.join(csrc, f) for f in os.listdir(csrc)) + [os.path.join(ROOT, 'include', 'pmg.h')]:
        with open(path, 'rb') as f:
            h.update(os.path.basename(path).encode() + b'\0' + f.read())
    return h.hexdigest()


def pmc_traffic(config, B, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic[_<config>].json, made by profiles/summarize_pmc.py; the counters cannot be
    read from inside a timed run).  None when the committed passes were taken at another configuration
    or batch, or on kernel sources other than the ones in the tree now (stale)."""
    try:
        name = 'pmc_traffic.json' if config == 'cfg2' else f'pmc_traffic_{config}.json'
        with open(os.path.join(ROOT, 'profiles', name)) as f:
            t = json.load(f)
        if t.get('config') != config or t.get('batch') != B:
            return None
        if t.get('kernel_sources_sha256') != kernel_source_hash():
            return None
        return t['kernels'][kernel]['total_GB'] * 1e9
    except (OSError, KeyError, ValueError):
        return None


def time_phases(learner, cfg, latent, rewards, B, reps=3):
    """Per-kernel durations with HIP events on the launch stream (torch's current stream)."""
    import ctypes as C
    from paccmann_generator_amd import _ffi, ops
    from paccmann_generator_amd.reinforce import pad_rows_on_device
    dec = learner.generator.decoder
    w = dec.weights()
    T = cfg['length']
    # batches beyond the workspace budget run as chunks: time the kernels on one chunk
    lo, hi = ops.plan_chunks(w.dims, B, T + 2)[0]
    if hi - lo < B:
        B = hi - lo
        latent, rewards = latent[:, :B].contiguous(), rewards[:B].contiguous()

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    out = {}
    tok_len = {}

    def f_dec():
        tok_len['v'] = ops.decode(w, latent, START, STOP, T, temperature=0.8, seed=1, offset=0, early_exit=False)
    out['decode'] = timed(f_dec)
    tokens, lens = tok_len['v']
    padded = pad_rows_on_device(tokens, lens, STOP, t_max=T + 2)
    Tt = padded.shape[0]
    with torch.no_grad():
        h0 = dec.latent_to_hidden(latent.repeat(dec.n_layers, 1, 1)).contiguous()
    shape = w.shape(B, Tt)
    ws = torch.empty(_ffi.workspace_bytes(shape, _ffi.PMG_MODE_TRAIN), dtype=torch.uint8, device=latent.device)
    wstruct, keep = w.struct(False)
    logp = torch.empty(Tt - 1, B, dtype=torch.float32, device=latent.device)
    grads = [torch.empty_like(p) for p in w.flat()]
    gstruct, keep2 = _ffi.pack_weight_struct(_ffi.PmgGrads, ops.DecoderWeights.unflatten(grads, dec.n_layers),
                                             dec.n_layers)
    gh0 = torch.empty_like(h0)
    loss = torch.empty(1, dtype=torch.float32, device=latent.device)
    st = _ffi.current_stream()

    def f_fwd():
        _ffi.check(_ffi.lib.pmg_logprob_fwd(C.byref(shape), C.byref(wstruct), _ffi._ptr(padded), _ffi._ptr(h0),
                                            _ffi._ptr(logp), None, ops._cluster_flag(), _ffi._ptr(ws), st), 'fwd')

    def f_bptt():
        _ffi.check(_ffi.lib.pmg_reinforce_bptt(C.byref(shape), C.byref(wstruct), _ffi._ptr(padded), _ffi._ptr(rewards),
                                               1.0 / B, ops._cluster_flag(), _ffi._ptr(gh0), _ffi._ptr(ws), st), 'bptt')

    def f_wgrad():
        _ffi.check(_ffi.lib.pmg_reinforce_wgrad(C.byref(shape), C.byref(wstruct), _ffi._ptr(padded), 0,
                                                C.byref(gstruct), _ffi._ptr(loss), _ffi._ptr(ws), st), 'wgrad')
    out['teacher_fwd'] = timed(f_fwd)
    out['bptt'] = timed(f_bptt)
    out['wgrad'] = timed(f_wgrad)
    out['_steps'] = dict(decode=T, teacher_fwd=Tt - 1, bptt=Tt - 1, wgrad=Tt - 1)
    out['_batch'] = B
    del ws
    return out


def _uses_cluster(learner, B, T):
    import ctypes as C
    from paccmann_generator_amd import _ffi, ops
    shape = learner.generator.decoder.weights().shape(B, T)
    return _ffi.lib.pmg_uses_cluster(C.byref(shape), _ffi.PMG_MODE_DECODE, ops._cluster_flag()) == 1


def _bptt_uses_cluster(learner, B, T):
    import ctypes as C
    from paccmann_generator_amd import _ffi, ops
    shape = learner.generator.decoder.weights().shape(B, T + 2)
    return (_ffi.lib.pmg_uses_cluster(C.byref(shape), _ffi.PMG_MODE_TRAIN, ops._cluster_flag()) & 2) != 0


def rank_command(n_gpus: int, argv, port: int):
    """Command line that starts the N ranks of `bench.py --gpus N` (one process per GPU on one node)."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n_gpus}',
            '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--config', default='cfg2', choices=sorted(CONFIGS))
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (default: the config\'s)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-phases', action='store_true')
    ap.add_argument('--no-cluster', action='store_true',
                    help='small batches: force the batch-partitioned forward kernels (A/B against the clustered ones)')
    ap.add_argument('--seqs', type=int, default=65536, help='cfg5: pre-sampled sequences per step (1M = 1048576)')
    ap.add_argument('--allow-variant', action='store_true',
                    help='permit a tuning build selected by PMG_LIB (ablation / profile variants give wrong or '
                         'slower results; the default library is the only one whose numbers count)')
    args = ap.parse_args()
    if os.environ.get('PMG_LIB') and not args.allow_variant:
        raise SystemExit('bench.py: PMG_LIB selects a non-default library; pass --allow-variant to time it')
    if os.environ.get('PMG_HANG_DUMP'):  # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['PMG_HANG_DUMP']), exit=True)

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # Called as plain `python bench.py --gpus N`: start the N ranks ourselves, one process per GPU,
        # as children of a process that has not touched the GPU (importing torch does not), and leave
        # with their exit code.  Rank 0's JSON line goes to our stdout.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        raise SystemExit(subprocess.call(rank_command(args.gpus, sys.argv[1:], port)))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}')
    import torch.distributed as dist
    pg = None
    if 'WORLD_SIZE' in os.environ and (world > 1 or os.environ.get('PMG_DIST_SINGLE')):
        # PMG_DIST_SINGLE=1 under a one-rank torch.distributed.run: the N>1 code path (process group,
        # RCCL all-reduce of the flat gradient buffer) with world size 1, for a box with one GPU
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # RCCL ("nccl" on ROCm) over xGMI.  PMG_DIST_BACKEND=gloo rehearses the N>1 code path on a
        # box with fewer GPUs than ranks (ranks then share devices; timing is meaningless there).
        backend = os.environ.get('PMG_DIST_BACKEND', 'nccl')
        local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        # device_id binds the communicator to this rank's GPU up front (no device guessing in barrier())
        dist.init_process_group(backend, **({'device_id': torch.device('cuda', local_rank)} if backend == 'nccl' else {}))
        pg = dist.group.WORLD
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)

    if args.no_cluster or world > max(1, torch.cuda.device_count()):
        # ranks that share a GPU (the gloo rehearsal on a box with fewer GPUs than ranks) must not both launch clustered
        # kernels: every member of a cluster has to be resident at once (include/pmg.h, PMG_FLAG_NO_CLUSTER)
        from paccmann_generator_amd import ops as _ops
        _ops.USE_CLUSTER = False
    cfg = dict(CONFIGS[args.config])
    B = args.batch or cfg['batch']
    T = cfg['length']
    learner = build_learner(cfg, device, pg)
    from paccmann_generator_amd import dp
    from paccmann_generator_amd.reinforce import pad_rows_on_device

    # synthetic, resident inputs
    torch.manual_seed(1 + rank)
    latent = torch.randn(1, B, cfg['latent_dim'], device=device)
    torch.manual_seed(2 + rank)
    rewards = torch.rand(B, device=device)
    valid_idx = torch.arange(B, device=device)

    def step():
        tokens, lens = learner.sample_tokens(latent, seq_index_base=rank * B, early_exit=False)
        padded = pad_rows_on_device(tokens, lens, STOP, t_max=T + 2)
        learner.learn(latent, padded, valid_idx, rewards, lrps=1, trusted_tokens=True)   # rows come straight from pmg_decode

    tokens_per_step = B * T
    if args.config == 'cfg5':
        # one step = one pass of teacher-forced log-probs + BPTT + weight gradients over args.seqs
        # pre-sampled sequences per GPU, in chunks of B whose gradients accumulate; one update at the end
        from paccmann_generator_amd import ops
        n_chunks = max(1, args.seqs // B)
        chunks = [presampled_tokens(B, T, cfg['vocab_size'], device, seed=5 + rank * 1000 + c) for c in range(min(n_chunks, 8))]
        dec = learner.generator.decoder
        tokens_per_step = n_chunks * B * T

        def step():  # noqa: F811
            learner.optimizer.zero_grad()
            for c in range(n_chunks):
                h0 = dec.latent_to_hidden(latent.repeat(dec.n_layers, 1, 1))
                loss, _ = ops.reinforce_loss(dec.weights(), chunks[c % len(chunks)], h0, rewards,
                                             1.0 / (n_chunks * B * world), trusted_tokens=True)
                loss.backward()
            if pg is not None:
                dp.all_reduce_gradients_(dec.parameters())
            torch.nn.utils.clip_grad_norm_(list(dec.parameters()), 0.1)
            learner.optimizer.step()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if pg is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if pg is not None:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if pg is not None:
        t = torch.tensor([el], device=device, dtype=torch.float64)
        dp.all_reduce_(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    replicas_identical = None
    if pg is not None:
        # replicas must still be identical after the steps (no broadcast is ever done)
        flat = torch.cat([p.detach().reshape(-1) for p in learner.generator.decoder.parameters()])
        lo, hi = flat.clone(), flat.clone()
        dp.all_reduce_(lo, op=dist.ReduceOp.MIN)
        dp.all_reduce_(hi, op=dist.ReduceOp.MAX)
        replicas_identical = bool(torch.equal(lo, hi))
        if not replicas_identical:
            raise SystemExit(f'rank {rank}: decoder replicas diverged')

    ms_per_step = 1e3 * el / args.steps
    tokens_per_s = world * tokens_per_step * args.steps / el
    F = flops_per_token(cfg)

    result = {
        'metric': ('SMILES tokens/sec (teacher-forced log-prob + BPTT step over pre-sampled sequences)'
                   if args.config == 'cfg5' else 'sampled SMILES tokens/sec (decode + REINFORCE step)'),
        'value': tokens_per_s, 'unit': 'tokens/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'{args.config}: {KIND[args.config]}{TAIL[args.config == "cfg5"]} hidden {cfg["rnn_cell_size"]}, '
                               f'stack {cfg["stack_depth"]}x{cfg["stack_width"]}, layers {cfg["n_layers"]}, '
                               f'vocab {cfg["vocab_size"]}, emb {cfg["embedding_size"]}, latent {cfg["latent_dim"]}',
                   'batch_per_gpu': B, 'global_batch': B * world, 'len': T, 'parallelism': f'dp{world}',
                   'flop_per_token_fwd': F},
        'step_roofline_frac': (tokens_per_s / world) * 3 * F / (PEAK_F32_MFMA_TFLOPS * 1e12),
        # the teacher-forced re-forward the reference performs is executed but not
        # counted in the 3F figure above; this is the executed (4F) figure alongside (cfg5 executes 3F)
        'step_executed_flop_frac': (tokens_per_s / world) * (3 if args.config == 'cfg5' else 4) * F /
                                   (PEAK_F32_MFMA_TFLOPS * 1e12),
        'tokens_per_step_per_gpu': tokens_per_step,
        'replicas_identical': replicas_identical,     # N>1: checked over all ranks after the timed steps
        'dist_backend': (dist.get_backend(pg) if pg is not None else None),
        'forward_kernels': 'clustered' if _uses_cluster(learner, B, T) else 'batch-partitioned',
        'bptt_kernels': 'clustered' if _bptt_uses_cluster(learner, B, T) else 'batch-partitioned',
    }
    if rank == 0:
        if not args.no_phases:
            ph = time_phases(learner, cfg, latent, rewards, B)
            steps = ph.pop('_steps')
            Bp = ph.pop('_batch')      # sequences per kernel launch (one chunk when the batch is chunked)
            dom = max(ph, key=lambda k: ph[k])
            ach = Bp * steps[dom] * F / (ph[dom] * 1e-3) / 1e12
            result['roofline'] = {'bound': 'mfma', 'kernel': dom, 'achieved': ach, 'peak': PEAK_F32_MFMA_TFLOPS,
                                  'unit': 'TFLOP/s', 'frac': ach / PEAK_F32_MFMA_TFLOPS,
                                  'traffic': pmc_traffic(args.config, Bp, dom), 'ms': ph[dom],
                                  'sequences_per_launch': Bp}
            result['phase_ms'] = ph
            result['phase_tflops'] = {k: Bp * steps[k] * F / (v * 1e-3) / 1e12 for k, v in ph.items()}
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(cfg)
        print(json.dumps(result), flush=True)
    if pg is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
