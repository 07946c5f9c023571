"""Collects VCF body lines that Pisces itself wrote (the reference's test-data VCFs) as golden vectors for the VCF formatter
(pisces_hip_format_vcf).  Usage: python tests/golden/extract_vcf_lines.py /root/reference  -> tests/golden/vcf_lines.json
Only data is kept: the body lines (one sample, FORMAT GT:GQ:AD:DP:VF[:NL:SB][:NC]) plus the q<N> / R<M>x<N> filter ids of the
file's header."""
import json
import os
import re
import sys

FILES = [
    "src/test/Scylla.Tests/TestData/small_S1.genome.vcf",
    "src/test/Scylla.Tests/TestData/chr21_11085587_S1.genome.vcf",
    "src/test/Scylla.Tests/TestData/Bcereus_S4.vcf",
    "src/test/Psara.Tests/TestData/PsaraTestInput.genome.vcf",
    "src/test/Pisces.IO.Tests/TestData/VcfReWriter_NoChangeToVariants.vcf",
]
FORMATS = {"GT:GQ:AD:DP:VF", "GT:GQ:AD:DP:VF:NL:SB", "GT:GQ:AD:DP:VF:NL:SB:NC", "GT:GQ:AD:DP:VF:NC"}


def main(root):
    out = []
    for rel in FILES:
        path = os.path.join(root, rel)
        q = rmxn = None
        source = ""
        lines = []
        with open(path) as f:
            for line in f:
                line = line.rstrip("\r\n")
                if line.startswith("##source="):
                    source = line[len("##source="):]
                m = re.match(r"##FILTER=<ID=q(\d+),", line)
                if m:
                    q = int(m.group(1))
                m = re.match(r"##FILTER=<ID=R(\d+)x(\d+),", line)
                if m:
                    rmxn = [int(m.group(1)), int(m.group(2))]
                if line.startswith("#") or not line:
                    continue
                cols = line.split("\t")
                if len(cols) != 10 or cols[8] not in FORMATS:
                    continue
                lines.append(line)
        out.append({"file": rel, "source": source, "q": q, "rmxn": rmxn, "lines": lines[:400]})
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vcf_lines.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=0)
    print({o["file"]: len(o["lines"]) for o in out})


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
