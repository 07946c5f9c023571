"""Golden vectors of forced genotyping (-forcedalleles): the VCF body lines Pisces left for ForcedGTFxnlTest.RunForcedGT
(src/test/Pisces.Tests/FunctionalTests/ForcedGTFxnlTest.cs:10-124) and the alleles of its input VCF.
    python tests/golden/extract_forced_gt.py /root/reference  ->  tests/golden/forced_gt.json
The reads and the genome of the test (SharedData/Bams/PhiX_S3.bam, Genomes/PhiX) are the ones of tests/golden/bam_phix.npz
(extract_bam_fixture.py).  Only data is kept: body lines and (position, ref, alt) triples."""
import json
import os
import sys

T = "src/test/Pisces.Tests/TestData"
RUNS = {   # expected file -> what the run differed in (the common options are in tests/test_oracle_golden.py)
    "noisy": ("PhiX_S3.noisy.vcf", dict(min_variant_qscore=1, forced=False)),
    "forced1": ("PhiX_S3.Forced1.vcf", dict(min_variant_qscore=1, forced=True)),
    "forced2": ("PhiX_S3.Forced2.vcf", dict(min_variant_qscore=20, forced=True)),
}


def body(path):
    with open(path) as f:
        return [l.rstrip("\r\n") for l in f if l.strip() and not l.startswith("#")]


def main(root):
    out = {"runs": {}}
    for name, (fn, what) in RUNS.items():
        out["runs"][name] = dict(what, file=T + "/" + fn, lines=body(os.path.join(root, T, fn)))
    forced = []
    for l in body(os.path.join(root, T, "PhiX_S3.forcedGTInput.vcf")):
        c = l.split("\t")
        ref, alt = c[3].upper(), c[4].upper()
        # Factory.GetForcedAlleles / IsValidAlt (src/exe/Pisces/Logic/Factory.cs:56-96): ref == alt and alts that are not all A/C/G/T are dropped
        if ref == alt or any(ch not in "ACGT" for ch in alt):
            continue
        forced.append([int(c[1]), ref, alt])
    out["forced"] = forced
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "forced_gt.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=0)
    print({k: len(v["lines"]) for k, v in out["runs"].items()}, "forced", forced)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
