"""Diploid (germline) known answers of the reference's tests as data:  python tests/golden/extract_diploid_cases.py /root/reference
 -> tests/golden/diploid_cases.json
  * genotype scenarios: every ExecuteDiploidGenotypeTest(...) call of src/test/Pisces.Genotyping.Tests/GenotypeCalculatorTest.cs:31-97
    (expected locus genotype, number of alleles to prune, reference / variant frequencies, coverage; harness :107-147: alleles at
    coverage 1000 unless stated, MinDepthToGenotype 100, thresholds 0.20 / 0.70 / 0.80);
  * genotype q-scores: the tables of DiploidGenotypeQualityCalculatorTests.cs:16-96,103-117 (genotype, depth, frequencies -> q-score);
  * PopulateDiploidStats: StrandBiasCalculatorTests.cs:185-285 (support, coverage -> FN, FP, P(var > 0) to three decimals; noise 0.01,
    threshold 0.20)."""
import json
import os
import re
import sys


def floats(s):
    return [float(x.rstrip("f")) for x in re.findall(r"[-+]?\d*\.?\d+f?", s)] if s.strip() else []


def main(root):
    t = os.path.join(root, "src/test")
    src = open(os.path.join(t, "Pisces.Genotyping.Tests/GenotypeCalculatorTest.cs"), encoding="utf-8-sig").read()
    body = src[src.index("public void DiploidGenotypeScenarios()"):src.index("private void ExecuteDiploidGenotypeTest(")]
    geno = []
    pat = re.compile(r"ExecuteDiploidGenotypeTest\(Genotype\.(\w+),\s*(\d+),\s*new List<float>\s*\{([^}]*)\},\s*new List<float>\s*\{([^}]*)\}"
                     r"(?:,\s*new List<FilterType>\s*\{[^}]*\},\s*(\d+))?\)")
    for m in pat.finditer(body):
        geno.append({"genotype": m.group(1), "prune": int(m.group(2)), "ref_freqs": floats(m.group(3)), "alt_freqs": floats(m.group(4)),
                     "coverage": int(m.group(5)) if m.group(5) else 1000})
    src = open(os.path.join(t, "Pisces.Genotyping.Tests/DiploidGenotypeQualityCalculatorTests.cs"), encoding="utf-8-sig").read()
    gq = []
    depth = None
    freqs = None
    exp = None
    for line in src.splitlines():
        m = re.search(r"depth = (\d+);", line)
        if m:
            depth = int(m.group(1))
        m = re.search(r"testFrequencies = new double\[\] \{([^}]*)\}", line)
        if m:
            freqs = [float(x) for x in re.findall(r"[\d.]+", m.group(1))]
        m = re.search(r"expectedResults = new int\[\] \{([^}]*)\}", line)
        if m:
            exp = [2147483647 if "MaxValue" in x else int(x) for x in re.findall(r"int\.MaxValue|\d+", m.group(1))]
        m = re.search(r"variant\.Genotype = Genotype\.(\w+);", line)
        if m:
            gq.append({"genotype": m.group(1), "depth": depth, "frequencies": freqs, "expected": exp})
    src = open(os.path.join(t, "Pisces.Calculators.Tests/UnitTests/StrandBiasCalculatorTests.cs"), encoding="utf-8-sig").read()
    body = src[src.index("StrandBiasStats stats = new StrandBiasStats(100, 100);"):]
    body = body[:body.index("public void")] if "public void" in body else body
    stats = []
    cur = None
    for line in body.splitlines():
        m = re.search(r"new StrandBiasStats\(([^;]*)\);", line)
        if m:
            args = m.group(1)
            k = args.rindex(",")
            sup_expr, cov = args[:k].strip().strip("()"), float(args[k + 1:])
            sup = eval(sup_expr)   # "15" or "20.0*0.15": arithmetic on literals only
            cur = {"support": sup, "coverage": cov}
            stats.append(cur)
        m = re.search(r"Assert\.Equal\(stats\.(\w+),\s*([\d.]+),\s*3\)", line)
        if m and cur is not None:
            cur[m.group(1)] = float(m.group(2))
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "diploid_cases.json")
    json.dump({"genotype_scenarios": geno, "genotype_qscores": gq, "diploid_sb_stats": stats, "thresholds": [0.20, 0.70, 0.80],
               "min_depth_to_genotype": 100, "sb_noise_freq": 0.01, "sb_threshold": 0.20}, open(dst, "w"), indent=1)
    print(len(geno), "genotype scenarios,", len(gq), "q-score tables,", len(stats), "stats cases")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
