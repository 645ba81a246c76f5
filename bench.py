#!/usr/bin/env python
"""Headline benchmark: frames/sec of 49x49 saliency maps, gaze_grcn, 16-step clips.

One step = one pass of the whole hot path over one batch of synthetic clips that is
already resident in HBM:

  e2e  (default)  video windows [B*T,16,112,112,3] fp32 -> C3D conv1a..conv5b ->
                  1024->512 projection -> ConvGRU (T steps) -> transposed-conv head ->
                  per-frame softmax  =>  B*T maps of 49x49
  head            c3d_input [B,T,1024,7,7] fp32 (what the reference's TF graph is fed,
                  models/gaze_rnn.py:118-121) -> the same head

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` ; for N>1 it is
launched by torch.distributed.run with one rank per GPU.  Clips shard data-parallel
across ranks with no data-path collective (inference: SURVEY.md 8e "replicas only"),
so scaling is weak: every rank runs B clips.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import threading
import time
import traceback

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from recurrent_gaze_prediction_amd import dist as rdist  # noqa: E402
from recurrent_gaze_prediction_amd import synthetic as syn  # noqa: E402
from recurrent_gaze_prediction_amd.engine import C3DEngine, GrcnEngine  # noqa: E402

# forward FLOPs (2*MAC), reference op sequence, no folding
C3D_LAYERS = [('conv1a', 3, 64, 16, 112), ('conv2a', 64, 128, 16, 56), ('conv3a', 128, 256, 8, 28),
              ('conv3b', 256, 256, 8, 28), ('conv4a', 256, 512, 4, 14), ('conv4b', 512, 512, 4, 14),
              ('conv5a', 512, 512, 2, 7), ('conv5b', 512, 512, 2, 7)]
C3D_FLOPS = {n: 2.0 * d * h * h * 27 * ci * co for n, ci, co, d, h in C3D_LAYERS}     # per window
C3D_NAMES = [l[0] for l in C3D_LAYERS]
HEAD_FLOPS = {'proj': 51.38e6, 'xconv': 173.41e6, 'convgru_seq': 43.35e6, 'head': 20.07e6 + 54.17e6 + 90.35e6 + 0.06e6}
HEAD_FLOPS_FRAME = 432.79e6
C3D_FLOPS_FRAME = sum(C3D_FLOPS.values())          # 76 993.27 MFLOP
PEAK_TFLOPS = {'bf16': 2500.0, 'f32': 157.3}       # dense MFMA peaks, MI355X_MICROARCH.md
PMC_SUMMARY = 'profiles/r05_pmc_summary.json'      # separate rocprofv3 --pmc passes of the default command (scripts/r05_profiles.sh)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', choices=['e2e', 'head', 'train', 'finetune'], default='e2e',
                    help="finetune = end-to-end training step incl. the conv stack's backward (BASELINE config 5 style); "
                         "train = head training step on precomputed features (BASELINE config 4: fwd + bwd + "
                         "gradient all-reduce + clipped Adam; pass --batch 8 --n-steps 35 for its shape)")
    ap.add_argument('--batch', type=int, default=64, help='clips per GPU')
    ap.add_argument('--n-steps', type=int, default=16, help='RNN timesteps T per clip')
    ap.add_argument('--dtype', choices=['bf16', 'f32'], default='bf16')
    ap.add_argument('--c3d-chunk', type=int, default=1024, help='windows per C3D launch chain')
    ap.add_argument('--graph', action='store_true', help='train workload: replay the step as HIP graphs')
    ap.add_argument('--dp-train-probe', choices=['auto', 'on', 'off'], default='auto',
                    help="after the headline timing also time BASELINE config 4's data-parallel training step (B=8 x T=35 per "
                         "GPU: fwd + bwd + bucketed RCCL all-reduce + clip + Adam) and the same step at config 3's per-GPU shape "
                         "(B=64 x T=16: full-chip persistent ConvGRU launches next to the collectives), reported under 'dp_train' / "
                         "'dp_train_b64'; auto = when more than one rank runs, so that the driver's N>1 command exercises the "
                         "collective path")
    ap.add_argument('--rehearse', action='store_true',
                    help="NOT a measurement: run the N>1 control flow on a ONE-GPU box -- gloo instead of RCCL, every rank on cuda:0, "
                         "ConvGRU as per-timestep launches (persistent launches of several processes must not share a device); "
                         "the line carries 'rehearsal': true.  tests/test_bench_gpu.py")
    ap.add_argument('--probe-timeout', type=int, default=int(os.environ.get('RGP_BENCH_PROBE_TIMEOUT_S', '420')),
                    help="seconds the N>1 data-parallel probes may take before rank 0 prints the headline line with "
                         "'dp_probe_error' and the process exits 3 (a collective that never returns must not cost the measurement)")
    ap.add_argument('--dp-finetune-probe', choices=['auto', 'on', 'off'], default='auto',
                    help="also time BASELINE config 5's data-parallel JOINT training step at its per-GPU shape (16 clips x T=35: "
                         "C3D + cascade, the gradient leaving in nine buckets on a side stream under the backward) with and "
                         "without the all-reduce, reported under 'dp_finetune'; auto = when more than one rank runs")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=15.0, help='CPU baseline budget')
    ap.add_argument('--cpu-threads', type=int, default=16, help='host threads for the CPU baseline')
    return ap.parse_args()


def cpu_baseline(args, budget_s):
    """Reference-equivalent CPU restatement (TF1.x unavailable offline): the oracle's
    torch-CPU fp32 graph, unfused and T-unrolled like the TF graph, on the host cores.
    Bounded sample: whole clips (T frames each) until ~budget_s seconds are spent."""
    from oracle import torch_ref
    # the GPU box gives one GPU's job a 16-core CPU share; more threads than that only
    # oversubscribe (measured: 256 threads ran the 7x7 convs 20x slower than 16)
    torch.set_num_threads(min(os.cpu_count() or 1, args.cpu_threads))
    T = args.n_steps
    hp = {k: torch.tensor(v) for k, v in syn.grcn_params(1, T).items()}
    cp = {k: torch.tensor(v) for k, v in syn.c3d_params(2).items()}
    frames, t_c3d, t_head = 0, 0.0, 0.0
    t_start = time.time()
    clip = 0
    with torch.no_grad():
        while True:
            if args.workload == 'e2e':
                v = torch.tensor(syn.video_windows(100 + clip, T))
                t0 = time.time()
                feat = torch_ref.c3d_forward(v, cp)                      # [T,1024,7,7]
                t_c3d += time.time() - t0
                x = feat.reshape(1, T, 1024, 7, 7)
            else:
                x = torch.tensor(syn.c3d_features(100 + clip, 1, T))
            t0 = time.time()
            torch_ref.softmax_maps(torch_ref.grcn_forward(x, hp))
            t_head += time.time() - t0
            frames += T
            clip += 1
            el = time.time() - t_start
            if el >= budget_s or el + el / clip > budget_s * 1.5:
                break
    tot = t_c3d + t_head
    return {'value': frames / tot, 'unit': 'frames/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '%d clip(s) of %d frames (%s), oracle/torch_ref.py fp32 on host cores, %.1f s '
                      '(C3D %.1f s, head %.2f s)' % (clip, T, args.workload, tot, t_c3d, t_head)}


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(rdist.spawn_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    rank, local_rank, world = rdist.env_world()
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device (the product path has no CPU fallback)')
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    control_note = None
    try:
        dist = rdist.init(backend='gloo' if args.rehearse else 'nccl', device=dev)      # 'nccl' is RCCL on ROCm; None when world == 1
    except Exception as e:                                       # noqa: BLE001
        # RCCL could not build its communicator (first N > 1 run on this pool): the headline has no data-path collective --
        # gloo can carry its barrier, MAX and per-rank times; the probes, which exist to exercise RCCL, are skipped and say so
        if args.rehearse or world == 1 or args.workload in ('train', 'finetune'):
            raise
        control_note = 'RCCL initialisation failed (%s: %s); gloo carries the barrier / MAX over ranks' % (type(e).__name__, str(e)[:300])
        sys.stderr.write('bench.py: %s\n' % control_note)
        import torch.distributed as tdist
        if tdist.is_initialized():                               # (a half-built default group)
            tdist.destroy_process_group()
        os.environ['MASTER_PORT'] = str(int(os.environ.get('MASTER_PORT', '29500')) + 1)   # the first store may still hold the port
        dist = rdist.init(backend='gloo', device=dev)
    B, T, F = args.batch, args.n_steps, args.batch * args.n_steps

    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    ft = None
    if args.workload == 'finetune':
        from recurrent_gaze_prediction_amd.finetune import EndToEndGaze
        ft = EndToEndGaze(B, T, dtype=args.dtype, device=dev, max_windows=min(args.c3d_chunk, F), seed=1, per_step=args.rehearse)
        ft.attach_process_group(dist)
        head, c3d = ft.head, ft.c3d
    else:
        head = GrcnEngine(B, T, dtype=args.dtype, save_for_backward=args.workload == 'train', device=dev, per_step=args.rehearse)
        head.set_weights(syn.grcn_params(1, T))
    logits = torch.empty(B, T, 49, 49, device=dev)
    probs = torch.empty_like(logits)
    if ft is not None:
        video = torch.rand(F, 16, 112, 112, 3, device=dev, generator=g) - 0.5
        gt = torch.rand(B, T, 49, 49, device=dev, generator=g) + 1e-3
        gt = (gt / gt.sum((-1, -2), keepdim=True)).contiguous()
        last = {}

        def step():
            k = ft.global_step
            last['loss'], last['gnorm'] = ft.train_step(video, gt, 1e-4 * 0.8 ** (k // 500), max_grad_norm=10.0)
    elif args.workload == 'e2e':
        c3d = C3DEngine(min(args.c3d_chunk, F), dtype=args.dtype, device=dev)
        c3d.set_weights(syn.c3d_params(2))
        video = torch.rand(F, 16, 112, 112, 3, device=dev, generator=g) - 0.5   # U(0,1)-0.5, resident in HBM
        rows = torch.empty(F * 49, 1024, dtype=c3d.torch_dtype, device=dev)

        def step():
            c3d.forward(video, want_features=False, want_rows=True, out_rows=rows)
            head.forward_rows(rows, out_logits=logits, out_probs=probs)
    elif args.workload == 'head':
        c3d = None
        x = torch.relu(torch.randn(B, T, 1024, 7, 7, device=dev, generator=g))    # conv5b is post-ReLU

        def step():
            head.forward(x, out_logits=logits, out_probs=probs)
    else:
        c3d = None
        x = torch.relu(torch.randn(B, T, 1024, 7, 7, device=dev, generator=g))
        gt = torch.rand(B, T, 49, 49, device=dev, generator=g) + 1e-3
        gt = (gt / gt.sum((-1, -2), keepdim=True)).contiguous()
        counter = [0]
        if args.graph:
            # the step is launch-bound at config 4's shape: replay it as two HIP graphs around the all-reduce
            from recurrent_gaze_prediction_amd.graph import GraphedHeadTrainStep
            gstep = GraphedHeadTrainStep(head, x, gt, 1e-4, 0.8, 500, 10.0, dist=dist)
            logits, probs = gstep.logits, gstep.probs
            step = gstep.step
        else:
            def step():
                head.forward(x, out_logits=logits, out_probs=probs)
                head.backward(logits, probs, gt)
                rdist.allreduce_mean_(dist, [head.flat_grads])        # RCCL, before the global-norm clip
                head.adam_step(counter[0], 1e-4 * 0.8 ** (counter[0] // 500), max_grad_norm=10.0)
                counter[0] += 1

    def barrier():
        rdist.barrier(dist, dev)

    for _ in range(args.warmup):
        step()
    barrier()
    head.profile(True)
    if c3d is not None:
        c3d.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    mine = time.perf_counter() - t0
    elapsed = rdist.max_over_ranks(dist, mine, dev)
    per_rank_ms = [round(t / args.steps * 1e3, 4) for t in rdist.gather_over_ranks(dist, mine, dev)]   # which GPU set the MAX
    hprof = head.profile_read()
    cprof = c3d.profile_read() if c3d is not None else {}
    if ft is not None:
        assert bool(torch.isfinite(last['loss'])) and bool(torch.isfinite(last['gnorm'])), 'non-finite training step'
    else:
        assert torch.isfinite(probs).all(), 'non-finite saliency maps'

    out = {}
    if rank == 0:
        frames_total = world * F * args.steps
        value = frames_total / elapsed
        # ---- roofline of the dominant kernel (HIP-event time on the launch stream, timed region)
        if c3d is not None:
            groups = {}
            for name, _, _, _, _ in C3D_LAYERS:
                ms, calls = cprof[name]
                # the library names the kernel it launched for this layer at this chunk size (rgp_c3d_layer_kernel_name)
                grp = groups.setdefault(c3d.layer_kernel_name(C3D_NAMES.index(name), min(args.c3d_chunk, F)), [0.0, 0.0, 0])
                grp[0] += ms
                grp[1] += C3D_FLOPS[name] * F * args.steps          # flops executed in the timed region
                grp[2] += calls
            kname, (ms, flops, calls) = max(groups.items(), key=lambda kv: kv[1][0])
        elif getattr(args, 'graph', False) and args.workload == 'train':
            # the stage timers live in the library calls, which a graph replay does not make: price the whole step
            kname, ms, calls = 'HIP-graph replay of the training step (fwd + bwd + clip + Adam)', elapsed * 1e3, args.steps
            flops = 3.0 * HEAD_FLOPS_FRAME * F * args.steps
        else:
            kname, (ms, calls) = max(((k, v) for k, v in hprof.items() if k != 'softmax'), key=lambda kv: kv[1][0])
            flops = HEAD_FLOPS[kname] * F * args.steps
        achieved = flops / (ms * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.dtype]
        # HBM bytes per launch of that kernel: PMC counters cannot be read from inside this process.  For the default
        # command the figure comes from the committed summary of separate `rocprofv3 --pmc` passes of this very command
        # (scripts/r03_profiles.sh -> scripts/pmc_summary.py: 2 x FETCH_SIZE + WRITE_SIZE, the guide's gfx950 correction;
        # Infinity-Cache hits are counted, so it is traffic beyond L2, an upper bound of HBM bytes); any other invocation
        # reports null.
        traffic, traffic_stale = None, None
        if c3d is not None and args.workload == 'e2e' and min(args.c3d_chunk, F) == 1024:
            try:
                from recurrent_gaze_prediction_amd import _lib as rlib
                with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), PMC_SUMMARY)) as fh:
                    pmc = json.load(fh)
                def stem_of(name):                                          # kernel<CIN,NOUT,HW,D (whatever follows the fourth number)
                    head, _, rest = name.replace(' ', '').partition('<')
                    return head + '<' + ','.join(rest.replace('>', ',').split(',')[:4])
                stem = stem_of(kname)
                hit = [v for k, v in pmc.items() if stem_of(k) == stem and 'hbm_bytes_per_launch' in v]
                # the counters belong to the build they were taken with: compare the sources of this kernel
                then = pmc.get('_meta', {}).get('kernel_source_hashes', {})
                now = rlib.kernel_source_hashes()
                files = rlib.KERNEL_SOURCES.get(kname.split('<')[0], tuple(now))
                traffic_stale = not then or any(then.get(f) != now.get(f) for f in files)
                if len(hit) == 1 and not traffic_stale:
                    traffic = round(hit[0]['hbm_bytes_per_launch'] / 1e9, 3)      # GB per launch
            except (OSError, ValueError):
                traffic, traffic_stale = None, True         # no summary of this round's build yet
        roofline = {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': peak, 'unit': 'TFLOP/s',
                    'frac': round(achieved / peak, 4), 'traffic': traffic, 'traffic_unit': 'GB per launch',
                    'traffic_profile': PMC_SUMMARY + ' (separate rocprofv3 --pmc passes of this command; null + traffic_stale '
                                       'when the kernel\'s sources have changed since)',
                    'traffic_stale': traffic_stale,
                    'kernel': kname,
                    'launches': int(calls), 'avg_launch_ms': round(ms / max(calls, 1), 4),
                    'algorithmic_gflop_per_launch': round(flops / max(calls, 1) / 1e9, 3)}
        if kname == 'head':
            roofline['note'] = ('algorithmic FLOPs of the three transposed convolutions + out_W (164.65 MFLOP per frame, SURVEY 8d); the '
                                'library runs them as their exact fold, which issues 4.8 MFLOP per frame: not an MFMA utilisation')
        flops_frame = HEAD_FLOPS_FRAME + (C3D_FLOPS_FRAME if c3d is not None else 0.0)
        if ft is not None:
            from recurrent_gaze_prediction_amd.finetune import flops_per_frame_train
            flops_frame = flops_per_frame_train()
        out = {
            'metric': 'frames/sec (49x49 saliency maps), gaze_grcn 16-frame clips',
            'value': round(value, 2), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 4),
            'per_rank_ms': per_rank_ms,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': ('gaze_grcn END-TO-END TRAINING step: synthetic 16x112x112x3 windows -> C3D conv1a-5b (arg-max '
                                    'recorded) -> gaze_grcn head -> xentropy loss -> head backward -> conv-stack backward '
                                    '(dgrad + wgrad of all 8 layers) -> gradient all-reduce -> global-norm clip + TF-Adam on '
                                    'all 30.7 M variables; roofline = dominant FORWARD kernel of the step'
                                    if ft is not None else
                                    'gaze_grcn end-to-end: synthetic 16x112x112x3 windows -> C3D conv1a-5b -> '
                                    '1024->512 proj -> ConvGRU(512->128, 7x7) -> deconv head -> 49x49 softmax maps'
                                    if c3d is not None else
                                    'gaze_grcn head on precomputed C3D conv5b features [B,T,1024,7,7]'
                                    if args.workload == 'head' else
                                    'gaze_grcn TRAINING step on precomputed features: forward + backward + gradient '
                                    'all-reduce (mean, before the clip) + clip_by_global_norm(10) + TF-Adam'),
                       'clips_per_gpu': B, 'n_lstm_steps': T, 'frames_per_step_per_gpu': F,
                       'parallelism': ('dp%d (clip-sharded, one flat RCCL all-reduce of the 12 MB gradient per step)' % world
                                       if args.workload == 'train' else
                                       'dp%d (clip-sharded, RCCL all-reduce of the 12 MB head + 110.6 MB conv gradient buckets per step)' % world
                                       if ft is not None else
                                       'dp%d (clip-sharded replicas, no collective)' % world),
                       'weights': 'random init (reference initialisers), GRU filters std 0.05'},
            'algorithmic_tflops': round(value * flops_frame / 1e12, 2),
            'stage_ms_per_step': {k: round(v[0] / args.steps, 4) for k, v in list(cprof.items()) + list(hprof.items())},
            'roofline': roofline,
        }
        if args.rehearse:
            out['rehearsal'] = True             # gloo, one device, per-step ConvGRU: control flow only, the numbers mean nothing
        if world == 1 and not args.no_cpu_baseline and args.workload in ('e2e', 'head'):
            out['cpu_baseline'] = cpu_baseline(args, args.cpu_seconds)
            out['speedup_vs_cpu_baseline'] = round(value / out['cpu_baseline']['value'], 1)

    # ---- N > 1: the driver's one command must also exercise the collective path (inference has none).  The probes run AFTER
    # the headline is measured and assembled, under a watchdog: whatever happens in them -- an exception on this rank, a peer
    # that died, a collective that never returns -- rank 0 still prints its ONE line (with 'dp_probe_error') and exits non-zero.
    emitted = threading.Lock()
    store = rdist.default_store() if dist is not None else None              # c10d's key-value store: a channel beside the collectives
    done = threading.Event()

    def emit(extra):
        if rank == 0 and emitted.acquire(blocking=False):
            out.update(extra)
            print(json.dumps(out), flush=True)
            if store is not None:
                store.set('rgp_probe_emitted', '1')

    def watchdog():
        # a peer that failed says so through the store (its collectives will never complete); otherwise the deadline
        t_end, why = time.monotonic() + args.probe_timeout, None
        while not done.wait(1.0):
            if store is not None and store.check(['rgp_probe_error']):
                why = store.get('rgp_probe_error').decode()
                break
            if time.monotonic() > t_end:
                why = 'the data-parallel probes did not finish within %d s' % args.probe_timeout
                break
        if why is not None:
            emit({'dp_probe_error': why + ' (the headline in this line is complete)'})
            sys.stderr.write('bench.py: rank %d gives up on the data-parallel probes: %s\n' % (rank, why))
            sys.stderr.flush()
            os._exit(3)

    want_train = args.dp_train_probe == 'on' or (args.dp_train_probe == 'auto' and world > 1)
    want_ft = args.dp_finetune_probe == 'on' or (args.dp_finetune_probe == 'auto' and world > 1)
    if control_note is not None:
        want_train = want_ft = False
        if rank == 0:
            out['control_plane'] = control_note
            out['dp_probe_error'] = 'skipped: ' + control_note
    probes, err = {}, None
    if (want_train or want_ft) and world > 1:
        threading.Thread(target=watchdog, daemon=True).start()
    try:
        inject = os.environ.get('RGP_BENCH_INJECT_PROBE_FAULT', '')           # tests: 'raise:<rank>' / 'hang:<rank>'
        if inject and world > 1 and int(inject.split(':')[1]) == rank:
            if inject.startswith('hang'):
                time.sleep(10 ** 6)
            raise RuntimeError('injected probe fault')
        if want_train:
            # BASELINE config 4's training step at its per-GPU shape ...
            probes['dp_train'] = rdist.dp_train_probe(dist, dev, rank=rank, batch=8, n_steps=35, steps=max(5, args.steps), warmup=2, dtype=args.dtype, per_step=args.rehearse)
            # ... and at config 3's per-GPU shape (64 clips x T = 16): the persistent ConvGRU forward and BPTT launches then take
            # all 256 CUs (one workgroup each), the one interaction with a live RCCL kernel no smaller shape shows; the probe
            # reports whether a launch lost a member and fell back ('convgru_fallbacks', 0 expected: the library releases no
            # gradient bucket ahead of a full-chip launch, include/rgp.h)
            probes['dp_train_b64'] = rdist.dp_train_probe(dist, dev, rank=rank, batch=64, n_steps=16, steps=max(5, args.steps), warmup=2, dtype=args.dtype,
                                                          per_step=args.rehearse)
        if want_ft:
            # ... and the one place the design overlaps communication with compute: config 5's joint step (122.7 MB of conv + head
            # gradient buckets + the cascade's 216 MB), timed with and without the collectives
            # (the headline's engines stay alive next to it: 25 GB + the probe's 45 GB of the 288 GB)
            probes['dp_finetune'] = rdist.dp_finetune_probe(dist, dev, rank=rank, batch=16, n_steps=35, steps=3, warmup=1, dtype=args.dtype,
                                                            per_step=args.rehearse)
    except Exception as e:                                                   # noqa: BLE001 -- reported, then the process fails
        err = '%s: %s' % (type(e).__name__, e)
        traceback.print_exc()
    if err is not None:
        # the peers may be inside a collective this rank will never join: no further collective from here.  Rank 0 prints; any
        # other rank tells rank 0 through the store and gives it a moment to print before the launcher tears the job down
        msg = 'rank %d: %s' % (rank, err)
        if rank == 0 or store is None:
            emit(dict(probes, dp_probe_error=msg))
        else:
            store.set('rgp_probe_error', msg)
            t_end = time.monotonic() + 30.0
            while time.monotonic() < t_end and not store.check(['rgp_probe_emitted']):
                time.sleep(0.5)
        os._exit(4)
    if dist is not None:
        dist.barrier()
    done.set()
    emit(probes)
    if dist is not None:
        dist.destroy_process_group()
    for name, probe in probes.items():
        if probe['ranks_seen'] != args.gpus or not probe['replicas_in_sync']:
            raise SystemExit('bench.py: %s saw %d ranks for --gpus %d (replicas in sync: %s)'
                             % (name, probe['ranks_seen'], args.gpus, probe['replicas_in_sync']))


if __name__ == '__main__':
    main()
