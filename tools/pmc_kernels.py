"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel family (sum of each counter).
usage: python tools/pmc_kernels.py <dir> [name-substring ...]"""
import collections
import csv
import glob
import sys

def main():
    d = sys.argv[1]
    keys = sys.argv[2:] or ["median15", "conv_mfma_f32_kernel<40", "conv_mfma_f32_kernel<36", "conv_mfma16_f32_kernel<16",
                            "conv_mfma16_f32_kernel<14", "canny_nms", "prep_rows4"]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for key in keys:
                if key in r["Kernel_Name"]:
                    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
                    n[(key, r["Counter_Name"])] += 1
    for k, v in agg.items():
        print(k)
        for c, x in sorted(v.items()):
            print("   %-28s %.5g   (%d dispatches)" % (c, x, n[(k, c)]))

if __name__ == "__main__":
    main()
