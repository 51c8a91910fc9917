"""Condenses rocprofv3 output (kernel-trace stats + the two PMC passes) into the small text /
JSON files kept under profiles/.  Usage:
    python tools/summarize_profiles.py <stats_dir> <pmc_fetch_dir> <pmc_write_dir> <out_prefix>
The PMC values are averaged per launch; FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in
KiB; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes (MI355X_MICROARCH.md, HBM
section), so the corrected read figure is twice the raw one."""
import collections, csv, glob, json, re, sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z_0-9:]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name


def pmc(d, counter):
    tot = collections.defaultdict(lambda: [0.0, 0])
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            tot[k][0] += float(r["Counter_Value"])
            tot[k][1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in tot.items()}


def main():
    stats_dir, fdir, wdir, out = sys.argv[1:5]
    rows = []
    for fn in glob.glob(stats_dir + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            rows.append((short(r["Name"]), int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"]),
                         float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
    rows.sort(key=lambda r: -r[3])
    f, w = pmc(fdir, "FETCH_SIZE"), pmc(wdir, "WRITE_SIZE")
    with open(out + "_kernel_summary.txt", "w") as o:
        o.write("rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-api --no-legs\n")
        o.write("%-34s %6s %10s %7s %9s %9s\n" % ("kernel", "calls", "avg_us", "pct", "min_us", "max_us"))
        for r in rows:
            o.write("%-34s %6d %10.2f %7.2f %9.2f %9.2f\n" % r)
        o.write("\nPMC passes (separate runs, --pmc FETCH_SIZE / --pmc WRITE_SIZE), KiB per launch, raw counter values;\n")
        o.write("read bytes corrected for gfx950 = 2 x FETCH_SIZE\n")
        o.write("%-34s %8s %14s %14s %16s\n" % ("kernel", "launches", "FETCH_SIZE_KiB", "WRITE_SIZE_KiB", "hbm_bytes_corr"))
        js = {}
        for k in sorted(f, key=lambda k: -f[k][0]):
            wv = w.get(k, (0.0, 0))[0]
            corr = (2 * f[k][0] + wv) * 1024
            js[k] = {"launches": f[k][1], "fetch_kib_raw": f[k][0], "write_kib_raw": wv, "hbm_bytes_corrected": corr}
            o.write("%-34s %8d %14.1f %14.1f %16.0f\n" % (k, f[k][1], f[k][0], wv, corr))
    json.dump(js, open(out + "_pmc_traffic.json", "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
