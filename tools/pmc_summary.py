"""Mean counter value per kernel from a rocprofv3 --pmc ... --output-format csv run (counter_collection.csv)."""
import csv, collections, glob, sys
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        acc[(row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(f"{k:60s} {c:12s} launches {len(v):4d}  mean {sum(v) / len(v):.6e}")
