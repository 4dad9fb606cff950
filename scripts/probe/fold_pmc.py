"""Print the mean of every counter per (kernel, grid) from rocprofv3 --pmc csv output directories: fold_pmc.py DIR..."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].replace("(anonymous namespace)::", "")[:70] + " grid " + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, {c: round(sum(x) / len(x)) for c, x in v.items()}, flush=True)
