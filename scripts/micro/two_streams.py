"""GPU box: do the memory-bound kernels of one mapping iteration (K1, dense backward, optimiser pass) overlap with the VALU-bound
tile kernel of ANOTHER when two independent loops are enqueued on two HIP streams?  Upper bound of what pipelining the view halves of
one iteration over two streams could give (that version would add cross-stream events on top).

    python scripts/micro/two_streams.py [--iters 60]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=60)
ap.add_argument("--scale-add", type=float, default=0.0)
a = ap.parse_args()
sys.argv = [sys.argv[0], "--no-extras", "--no-cpu-baseline", "--refine-iters", "0"]
B = bench.Bench(bench.parse())
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
loops = []
for k, s in enumerate((s1, s2)):
    with torch.cuda.stream(s):
        loop, cams = B.build("fused", a.scale_add, seed_shift=k)
        B.run_steps(loop, 120)
    loops.append(loop)
torch.cuda.synchronize()


def run(conc, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if conc:            # one host thread per loop (a map() call ends with a host-side check of its stream)
            import threading

            def work(loop, s):
                with torch.cuda.stream(s):
                    B.run_steps(loop, a.iters)
                    s.synchronize()
            th = [threading.Thread(target=work, args=(loop, s)) for loop, s in zip(loops, (s1, s2))]
            for t in th:
                t.start()
            for t in th:
                t.join()
        else:
            with torch.cuda.stream(s1):
                for loop in loops:
                    B.run_steps(loop, a.iters)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


seq, conc = run(False), run(True)
print(json.dumps({"iters_per_loop": a.iters, "sequential_ms_per_iteration": round(1e3 * seq / (2 * a.iters), 4),
                  "two_streams_ms_per_iteration": round(1e3 * conc / (2 * a.iters), 4), "ratio": round(conc / seq, 3)}))
