"""A/B timing of EDT builds from several library builds, interleaved so that drift of the box cancels.

    python tools/edt_ab.py build_ab/A build_ab/B [--rounds 5]

Each directory holds a libicpx.so; every round runs the EDT-only bench once per directory (fresh process).
Prints mean and spread of ms per build and of the per-kernel times."""
import json, os, subprocess, sys, statistics

def main():
    args = sys.argv[1:]
    rounds = 5
    if "--rounds" in args:
        i = args.index("--rounds"); rounds = int(args[i + 1]); del args[i:i + 2]
    dirs = args
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {d: [] for d in dirs}
    for r in range(rounds):
        for d in dirs:
            env = dict(os.environ, ICPX_LIB_DIR=os.path.join(root, d))
            out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-extra", "--no-cpu", "--steps", "200", "--warmup", "20"],
                                 env=env, capture_output=True, text=True, timeout=300)
            if out.returncode != 0:
                print(out.stderr[-2000:]); sys.exit(1)
            j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
            res[d].append((j["ms_per_step"] * 1e3, j["roofline"]["stage_ms"]["pass_zy"] * 1e3, j["roofline"]["stage_ms"]["pass_x"] * 1e3))
            print(r, d, ["%.1f" % v for v in res[d][-1]], flush=True)
    for d in dirs:
        cols = list(zip(*res[d]))
        print(d, "build/zy/x us: " + "  ".join("%.1f +- %.1f (min %.1f)" % (statistics.mean(c), statistics.pstdev(c), min(c)) for c in cols))

if __name__ == "__main__":
    main()
