"""Aggregate rocprofv3 --pmc counter_collection CSVs: mean counter value per launch, per kernel.
usage: pmc_any.py <dir> [kernel-substring]"""
import collections, csv, glob, re, sys

tot, cnt = collections.defaultdict(float), collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if len(sys.argv) > 2 and sys.argv[2] not in k:
            continue
        tot[(k, r["Counter_Name"])] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for (k, c) in sorted(tot):
    print("%-50s %-28s n=%3d  mean %.4g" % (k[:50], c, cnt[(k, c)], tot[(k, c)] / cnt[(k, c)]))
