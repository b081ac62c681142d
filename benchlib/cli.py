"""bench.py's command line."""
import argparse
import os

from .common import ROOT


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--log2n", type=int, default=None, help="gates per GPU per step (default 2^20, the metric's batch; with 8 ranks 2^21 = "
                    "BASELINE config 3's 2^24 gates over 8 GPUs; steps above 2^20 gates run as 2^20-gate ranges, so the per-gate work is identical)")
    ap.add_argument("--layout", choices=["aos", "split"], default="split",
                    help="HBM layout of share vectors: arkworks AoS (drop-in) or engine-native split columns")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-log2n", type=int, default=20, help="CPU baseline sample size (gates)")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--settle-ms", type=float, default=30.0, help="untimed: run the pipeline for this long BEFORE the W warm-up steps, on every rank, so that the "
                    "timed region does not sit inside the power controller's transient after idle (probes/ramp_probe.py: a cold MI355X runs the first steps fast, "
                    "then creeps from 0.195 to 0.21-0.25 ms per step for several ms before settling at 0.198).  0 disables.  Reported in config.settle_ms.")
    ap.add_argument("--single-process", action="store_true", help="N GPUs driven by ONE process through the C ABI's multi-device group (arkmpc_group_*): what a "
                    "Rust party, which is one process, would run.  The driver's N>1 runs use torch.distributed.run (one process per GPU); this mode is the same "
                    "sharding behind the FFI.  Reports ranks_seen, per-member kernel times and the peer-write gather rate.")
    ap.add_argument("--devices", default=None, help="--single-process: comma-separated device ids of the members (default 0..N-1); ids may repeat "
                    "(members then share a GPU: how the mode is exercised on a one-GPU box)")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold pass (same W + K region with --settle-ms 0, run first) reported as value_cold / frac_cold")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra legs reported next to the headline at N=1 (AoS layout, config 4, config 5)")
    ap.add_argument("--min-timed-ms", type=float, default=50.0, help="the timed region repeats the K steps in whole rounds until it is at least this long "
                    "(K = 20 steps of 0.19 ms would be a 3.8 ms region: too short for the driver's clock and the power controller); 0 = exactly K steps. "
                    "Reported: steps = K, timed_rounds, timed_steps_total, timed_region_ms; ms_per_step and value are over the whole region")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="weak: --log2n gates per GPU per step whatever N (the default, what the driver runs); "
                    "strong: 2^--total-log2n gates per step in total (BASELINE config 3: 2^24), cut into N contiguous ranges")
    ap.add_argument("--total-log2n", type=int, default=24, help="--scaling strong: total gates per step over all GPUs")
    ap.add_argument("--only-e2e", action="store_true", help="run only the end-to-end (host records in, host records out) leg and print its JSON")
    ap.add_argument("--e2e-log2n", type=int, default=20, help="gates per party of the end-to-end leg")
    ap.add_argument("--only-circuit", action="store_true", help="run only the circuit leg (resident operands, triples from host memory) and print its JSON")
    ap.add_argument("--circuit-log2n", type=int, default=20, help="gates per batch_mul of the circuit leg")
    ap.add_argument("--circuit-depth", type=int, default=8, help="dependent gates in the circuit leg's chain")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the timed ordered all-gather of opened-value buffers (config 5 shape, 64 MiB per rank)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only for launch-path tests)")
    ap.add_argument("--sets", type=int, default=2, help="independent workload sets rotated step by step, so that no input line of step s "
                    "can still be cached (256 MiB Infinity Cache) when step s+1 runs; 1 = reuse the same buffers every step")
    ap.add_argument("--event-every", type=int, default=4, help="time the four kernels of every k-th timed step with dispatch-bound HIP events (at most 16 steps); "
                    "the whole region is bracketed by one event pair regardless")
    ap.add_argument("--k3-order", default="01", choices=["01", "10"], help="order of the two parties' K2+K3 launches after K1(P0), K1(P1). "
                    "The parties are independent; measured: no difference (within +-1 %).")
    ap.add_argument("--chunks", type=int, default=0, help="split each step's batch into this many gate ranges, each run K1,K1,K3,K3. "
                    "0 = automatic: ranges of 2^20 gates, so that the d||e buffers both parties exchange (128 MiB per range) stay in "
                    "the 256 MiB Infinity Cache between K1 and K3 -- measured: 2^21 gates/step 4.6e9 -> 5.1e9 gates/s, 2^22: 4.7e9 -> 5.2e9; "
                    "smaller ranges lose (2^20 in two halves: 4.6e9)")
    ap.add_argument("--detail-file", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"), help="where rank 0 writes everything that is not on the "
                    "compact headline line: every leg's full record, notes, per-kernel figures (one JSON object).  '' = do not write")
    ap.add_argument("--legs-to-stderr", action="store_true", help="also print every leg's record to stderr as a {\"leg\": name, ...} line")
    return ap.parse_args()
