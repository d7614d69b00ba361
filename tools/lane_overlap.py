#!/usr/bin/env python
"""Do the two lanes of the tensor-parallel prefill really run side by side?  Run under `rocprofv3 --kernel-trace --output-format csv`:
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python tools/lane_overlap.py run [tp] [S]
    python tools/lane_overlap.py read <dir>
`run` issues (after warm-up, separated by 30 ms pauses) a serial and a two-lane prefill of rank 0's shard with eager launches, then the
same two replayed from hipGraphs (1-rank RCCL communicator); `read` cuts the kernel trace at the pauses and reports per segment: kernels, span, sum of the kernel
durations, and the share of the span during which >= 2 kernels were executing (0 for a single stream).
Finding (profiles/r05_tp_prefill_two_lane_kernel_trace.log): rocprofv3's kernel trace serialises the dispatches, graph replay included, so
the overlap itself is not visible here -- the trace gives the SUM of the kernel durations (two-lane 30.3 ms vs serial 21.8 ms), against
which the unprofiled graph replay (23.8 ms) shows >= 6.5 ms of kernels running side by side; eager launches get none (host-bound)."""
import csv, glob, os, sys, time


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from emu_amd import synth
    from emu_amd.conf.emu_conf import LlamaCfg
    from emu_amd.llama import EmuHipContext, LlamaEngine
    tp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 1544
    dev = torch.device("cuda", 0)
    real = EmuHipContext(dev, 0, 1)
    real.init_tp(lambda b: b, force=True)

    class ShardView:
        def __init__(self, ctx, size):
            self.__dict__.update(ctx=ctx, tp_rank=0, tp_size=size)

        def __getattr__(self, k):
            return getattr(self.ctx, k)
    l = LlamaCfg()
    eng = LlamaEngine(l, 32274, ShardView(real, tp))
    eng.load_weights(synth.iter_synth(synth.llama_param_shapes(l, 32274), seed=0, device=dev, dtype=torch.bfloat16))
    x = (torch.randn(1, S, l.hidden_size, device=dev) * 0.1).to(torch.bfloat16)
    mask = torch.ones(1, S, dtype=torch.long, device=dev)
    with torch.no_grad():
        cap = eng.kv_capacity(S + 64)
        for rows in (0, 1024):
            eng.set_tp_overlap(rows)
            eng.prefill(x, mask, cap)
        torch.cuda.synchronize()
        graphs = []
        for rows in (0, 1024):                                   # replayed from hipGraphs: no host enqueue between the launches
            eng.set_tp_overlap(rows)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                eng.prefill(x, mask, cap)
            graphs.append(g)
        torch.cuda.synchronize()
        for rows in (0, 1024):
            time.sleep(0.03)
            eng.set_tp_overlap(rows)
            eng.prefill(x, mask, cap)
            torch.cuda.synchronize()
        for g in graphs:
            time.sleep(0.03)
            g.replay()
            torch.cuda.synchronize()


def read(d):
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    segs, cur = [], []
    for r in rows:
        if cur and r[0] - max(e for _, e, _ in cur) > 20_000_000:
            segs.append(cur); cur = []
        cur.append(r)
    if cur:
        segs.append(cur)
    print(f"{len(rows)} kernels, {len(segs)} segments (cut at pauses > 20 ms); the last four are: serial eager, two-lane eager, serial graph replay, two-lane graph replay")
    for i, s in enumerate(segs[-4:]):
        ev = sorted([(a, 1) for a, _, _ in s] + [(b, -1) for _, b, _ in s])
        t0, t1 = s[0][0], max(e for _, e, _ in s)
        depth, last, busy2, busy1 = 0, t0, 0, 0
        for t, dlt in ev:
            if depth >= 2: busy2 += t - last
            if depth >= 1: busy1 += t - last
            depth += dlt; last = t
        tot = sum(b - a for a, b, _ in s)
        print(f"segment {i}: {len(s)} kernels, span {(t1 - t0) / 1e6:.2f} ms, sum of kernel durations {tot / 1e6:.2f} ms, "
              f">= 1 kernel executing {busy1 / (t1 - t0):.2f} of the span, >= 2 kernels executing {busy2 / (t1 - t0):.2f}")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else read(sys.argv[2])
