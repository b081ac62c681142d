#!/usr/bin/env python3
"""bench.py -- authenticated Beaver mul-gates/sec over BN254 Fr, batch 2^20 per GPU (BASELINE.json config[1]).

A "step" is one complete two-party `AuthenticatedScalarResult::batch_mul` (reference:
online-phase/src/algebra/scalar/authenticated_scalar.rs:848-879) over a batch of n gates, both
parties simulated on the same GPU with the network mocked as a pointer swap (as
online-phase/benches/batch_ops.rs:20-39 runs both parties in-process):

    party0.K1 beaver_mask -> party1.K1 beaver_mask -> [exchange d||e] ->
    party0.K2+K3 beaver_finish_fused -> party1.K2+K3 beaver_finish_fused        (4 kernel launches)

All inputs are resident in HBM before the timed region.  Data is synthetic: uniformly random field
elements, valid Beaver triples and valid SPDZ MACs generated on the GPU from a fixed seed.

Launch:  python bench.py [--gpus N --steps K --warmup W]         (N = 1)
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
Multi-GPU = the same batch size on every rank (weak scaling), gates are independent so there is no
data-path collective; ranks only meet at the timing barriers.

Output contract: rank 0 ends stdout with ONE compact JSON line (budget 4 KB, hard cap 8 KB: numbers
and short identifiers only) -- metric, value, ms_per_step, config, dtype, roofline, cpu_baseline, the
result checks and one scalar or two per extra leg.  Everything else (every leg's full record, the
notes, per-kernel figures) goes to --detail-file (default gpurun_out/bench_detail.json).  The legs
live in benchlib/; each returns (summary, detail, ok).
"""
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib.cli import parse                                   # noqa: E402
from benchlib.common import FID                                  # noqa: E402

LINE_BUDGET, LINE_HARD_CAP = 4096, 8192


def init_ranks(args):
    """one process per GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torch.distributed.run); RCCL unless --dist-backend says otherwise"""
    from benchlib.headline import Ranks
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:     # under torch.distributed.run the collective path is used even for one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local_rank = local_rank % torch.cuda.device_count()      # (tests may oversubscribe one GPU with the gloo backend)
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)
    else:
        dist = None
        torch.cuda.set_device(0)
    return Ranks(dist, world, rank, local_rank, args.dist_backend)


def only_leg(fn, eng):
    """--only-e2e / --only-circuit: one leg alone, its full record on stdout (diagnostic modes, not what the driver runs)"""
    _, detail, ok = fn()
    print(json.dumps(detail), flush=True)
    eng.close()
    if not ok:
        raise SystemExit("result check failed")


def emit(line, detail, args):
    """the detail file first (never allowed to fail the run), then the ONE stdout line, trimmed of optional scalars if it ever outgrew the cap"""
    detail = dict(detail, headline=line)
    if args.detail_file:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(args.detail_file)), exist_ok=True)
            with open(args.detail_file, "w") as f:
                json.dump(detail, f)
            line["detail_file"] = os.path.relpath(args.detail_file, ROOT)
        except OSError as ex:
            print("bench.py: detail file not written: %r" % (ex,), file=sys.stderr)
    if args.legs_to_stderr:
        for k, v in detail.items():
            if isinstance(v, dict) and k != "headline":
                print(json.dumps(dict({"leg": k}, **v)), file=sys.stderr)
    s = json.dumps(line)
    for k in ("legs", "per_rank_ms_per_step", "rank_devices", "value_cold"):       # (does not happen at 1..8 ranks: the line is ~2.5 KB)
        if len(s) <= LINE_HARD_CAP:
            break
        line.pop(k, None)
        s = json.dumps(line)
    sys.stderr.flush()
    print(s, flush=True)


def main():
    args = parse()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    if args.single_process:
        from benchlib.group import main_single_process
        return main_single_process(args)
    ranks = init_ranks(args)
    world, rank, dist = ranks.world, ranks.rank, ranks.dist
    dev = torch.cuda.current_device()
    pkg = importlib.import_module("ark-mpc_amd")
    eng = pkg.Engine(FID, device=dev, host_buffers=False, stream=torch.cuda.current_stream().cuda_stream)
    if args.only_circuit:
        from benchlib.circuit import leg_circuit
        return only_leg(lambda: leg_circuit(pkg, eng, dev, args.circuit_log2n, args.circuit_depth), eng)
    if args.only_e2e:
        from benchlib.e2e import leg_end_to_end
        return only_leg(lambda: leg_end_to_end(pkg, eng, dev, args.e2e_log2n), eng)
    if args.scaling == "strong":                   # fixed total work: 2^total_log2n gates per step cut into `world` contiguous ranges
        if (1 << args.total_log2n) % world:
            raise SystemExit("--scaling strong needs a power-of-two number of GPUs")
        n = (1 << args.total_log2n) // world
        args.log2n = int(np.log2(n))
    else:
        if args.log2n is None:
            args.log2n = 21 if world == 8 else 20      # 8 ranks: BASELINE config 3 (2^24 gates over 8 GPUs)
        n = 1 << args.log2n

    from benchlib.headline import run_headline
    line, detail, ok, sets, chunks = run_headline(args, eng, ranks, n)
    gather = None
    if dist is not None and world > 1 and not args.no_gather:      # every rank takes part
        from benchlib.group import leg_gather
        gather = leg_gather(dist, world, rank, args.dist_backend)
    if rank == 0:
        if gather is not None:
            detail["gather"] = gather
            line["gather_ms"], line["gather_bus_GBps_per_rank"], line["gather_ordered"] = gather["ms"], gather["bus_GBps_per_rank"], gather["ordered"]
        if not args.no_cpu_baseline and world == 1:
            from benchlib.cpu import cpu_baseline
            from benchlib.pipeline import oracle_bitexact
            parties = sets[0][0]
            line["cpu_baseline"], detail["cpu_baseline"], res, myde, m = cpu_baseline(parties, n, args.cpu_log2n, args.layout)
            # the oracle's result for the same seeded workload is the bit-exact expectation for the buffers the timed region wrote
            exact = oracle_bitexact(parties, n, m, chunks, args.layout, res, myde)
            line["oracle_bitexact_gates"], line["oracle_bitexact_of"] = exact, m
            detail["oracle_bitexact_note"] = ("both parties' d||e and result records of workload set 0 after the timed region vs oracle/ark_oracle.c, every word, "
                                              "%d of %d gates compared" % (m, n))
            ok = ok and exact == m
            del parties, res, myde
        if world == 1 and not args.no_extras:
            del sets
            torch.cuda.empty_cache()
            from benchlib.aos import leg_aos
            from benchlib.circuit import leg_circuit
            from benchlib.config4 import leg_config4
            from benchlib.config5 import leg_config5
            from benchlib.e2e import leg_end_to_end
            legs = (("end_to_end", lambda: leg_end_to_end(pkg, eng, dev, args.e2e_log2n)),
                    ("circuit", lambda: leg_circuit(pkg, eng, dev, args.circuit_log2n, args.circuit_depth)),
                    ("aos", lambda: leg_aos(eng, n, args)),
                    ("config4", lambda: leg_config4(eng)),
                    ("config5", lambda: leg_config5(pkg, dev)))
            line["legs"] = {}
            for name, fn in legs:
                summary, detail[name], ok_leg = fn()
                line.update(summary)                 # the figures a reader should not have to dig for, next to `value`
                line["legs"][name] = "ok" if ok_leg else "FAILED"
                ok = ok and ok_leg
                torch.cuda.empty_cache()
        if not ok:
            line["results_check"] = line["results_check"].rsplit(":", 1)[0] + ": FAILED"
        emit(line, detail, args)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()
    if not ok:
        raise SystemExit("result check failed")


if __name__ == "__main__":
    main()
