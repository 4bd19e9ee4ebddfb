#!/usr/bin/env python3
"""HBM traffic per launch of the dominant pass kernel from two rocprofv3 --pmc runs of bench.py (FETCH_SIZE, WRITE_SIZE;
separate passes, --kernel-trace only): writes the JSON that bench.py reports as roofline.traffic.
   python tools/pmc_traffic.py <dir with pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/> <out.json> [keys_per_launch]
gfx950 correction (MI355X_MICROARCH.md, HBM section; calibrated here against a known 256-MiB streaming read):
FETCH_SIZE counts half the bytes of a coalesced streaming read -> x2; both counters are in KiB."""
import csv, glob, json, os, sys
root, out = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 26
# the dominant pass kernel: the MSD bucket pass when the run used the large keys-only sort, else the one-sweep pass
match, profile_name = ("onesweep_chain_kernel", "unsigned int>, 8"), "onesweep_u32_8b"
for f in glob.glob(os.path.join(root, "pmc_FETCH_SIZE", "**", "*counter_collection.csv"), recursive=True):
    if "msd_bucket_scatter_kernel<unsigned int, 512, 32, 1>" in open(f).read():   # pass 1 of the large sort (round 3: a kernel name of its own)
        match, profile_name = ("msd_bucket_scatter_kernel<unsigned int, 512, 32, 1>", ""), "msd2_pass1_u32"
vals = {}
kname = None
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    tot, cnt = 0.0, 0
    for f in glob.glob(os.path.join(root, "pmc_" + ctr, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "")
                if match[0] in k and match[1] in k and row.get("Counter_Name") == ctr:
                    tot += float(row["Counter_Value"]); cnt += 1
                    kname = k.split("(")[0].replace("void adlhip::", "")
    if cnt == 0:
        sys.exit("no %s rows for the pass kernel under %s" % (ctr, root))
    vals[ctr] = (tot / cnt, cnt)
fetch_kb, nf = vals["FETCH_SIZE"]; write_kb, nw = vals["WRITE_SIZE"]
traffic = int(fetch_kb * 1024 * 2.0 + write_kb * 1024)
json.dump({
    "kernel": kname, "profile_name": profile_name, "keys_per_launch": n, "launches_averaged": min(nf, nw),
    "FETCH_SIZE_kb": round(fetch_kb, 1), "WRITE_SIZE_kb": round(write_kb, 1), "fetch_correction": 2.0,
    "correction_note": "gfx950: FETCH_SIZE reports half of the bytes of a coalesced streaming read (MI355X_MICROARCH.md, HBM) -> x2; the "
                       "guide calibrates that for 16-B-per-lane loads, this kernel loads one dword per lane: 2 x FETCH_SIZE here equals "
                       "the %.1f MB of keys (+ ~4 MB of status rows in the one-sweep pass) the kernel is known to read, so the factor holds for it; both counters "
                       "sit on the memory side of L2 (fabric requests), Infinity-Cache hits included" % (n * 4 / 1e6),
    "traffic_bytes_per_launch": traffic,
    "algorithmic_bytes_per_launch": 2 * n * 4,
    "how": "two separate runs of `rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python3 bench.py --steps 5 --warmup 1 "
           "--no-cpu-baseline --no-verify` (tools/gpu_session.sh)",
}, open(out, "w"), indent=1)
print(open(out).read())
