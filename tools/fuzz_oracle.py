"""Runs tests/fuzz_cases.py over a range of seeds and prints the ones whose records differ from the oracle's.
    python tools/fuzz_oracle.py [first_seed] [n_seeds]
"""
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.fuzz_cases import one, one_tuples   # noqa: E402

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "tuples":   # the device-resident surface: python tools/fuzz_oracle.py tuples first n
        bad = 0
        first, n = int(sys.argv[2]), int(sys.argv[3])
        for seed in range(first, first + n):
            try:
                why, kw, rows = one_tuples(seed, verbose=True)
            except Exception as e:   # noqa: BLE001
                why, kw, rows = "raised: " + repr(e)[:300], None, 0
                traceback.print_exc(limit=3)
            print(("SEED %d DIFFERS: %s %s" % (seed, why, kw)) if why else ("seed %d ok %d" % (seed, rows)), flush=True)
            bad += bool(why)
        print("differing seeds:", bad, "of", n)
        sys.exit(0)
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    bad = 0
    for seed in range(first, first + n):
        try:
            why, kw, rows, forced = one(seed, verbose=True)
        except Exception as e:   # noqa: BLE001
            why, kw, rows, forced = "raised: " + repr(e)[:300], None, 0, None
            traceback.print_exc(limit=3)
        if why:
            bad += 1
            print("SEED", seed, "DIFFERS:", why, kw, forced, flush=True)
        else:
            print("seed", seed, "ok", rows, flush=True)
    print("differing seeds:", bad, "of", n)
