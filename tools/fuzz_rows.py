"""python tools/fuzz_rows.py seed position: the product's and the oracle's rows of a fuzz draw around a position."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.fuzz_cases import one   # noqa: E402

seed, pos = int(sys.argv[1]), int(sys.argv[2])
why, kw, (got, alleles, exp, exp_alleles, intervals, form), forced = one(seed, rows_too=True)
print(why, form[0], kw)
print("intervals near:", [iv for iv in (intervals or []) if iv[1] >= pos - 150 and iv[0] <= pos + 150])
for name, rows, al in (("product", got, alleles), ("oracle", exp, exp_alleles)):
    print(name)
    for r, a in zip(rows, al):
        if abs(int(r["position"]) - pos) <= 6 or (len(a[0]) > 1 and abs(int(r["position"]) - pos) < 12):
            print("  ", int(r["position"]), a, "cov", int(r["total_coverage"]), "sup", int(r["allele_support"]), "refsup", int(r["reference_support"]), "vq", int(r["variant_qscore"]),
                  "filters", hex(int(r["filter_bits"])))
