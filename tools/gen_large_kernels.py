#!/usr/bin/env python3
"""Regenerates oclradixsort_amd/csrc/{finish,wavefinish,passes,perdigit}_kernels.inc from the kernel symbols of a full build of libadlhip.so: the
instantiations of the kernel families that kernels_finish.hip / kernels_passes.hip / kernels_perdigit.hip compile in translation units of their own.
   make -C oclradixsort_amd/csrc single && python tools/gen_large_kernels.py"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "oclradixsort_amd", "lib", "libadlhip.so")
FAMILIES = {
    "finish": "wg_segment_sort_kernel|bin_segment_sort_kernel|segment_sort_kernel|wave_finish16_kernel",
    "wavefinish": "wave_segment_sort_kernel",
    "passes": "msd_bucket_scatter_kernel|msd_scatter_persist_kernel|msd_lookback_scatter_kernel|msd2s_offsets_kernel|msd2_offsets_kernel",
    "perdigit": "onesweep_chain_kernel|radix_scatter_kernel|onesweep_hist_kernel|small_sort_kernel|radix_count_kernel",
}
# (the demangler spells an ext_vector_type its own way)
stubs = [re.sub(r"^[0-9a-f]+ . ", "", line).replace("__device_stub__", "").replace("unsigned int __vector(4)", "adlhip::u32x4").strip()
         for line in subprocess.run(["nm", "-C", "--defined-only", so], capture_output=True, text=True, check=True).stdout.splitlines()
         if "__device_stub__" in line]
for name, fam in FAMILIES.items():
    sigs = sorted({s for s in stubs if re.search(r"adlhip::(%s)<" % fam, s)})
    head = ["// %s_kernels.inc -- GENERATED (tools/gen_large_kernels.py) from the kernel symbols of a full build: the kernel-template instantiations" % name,
            "// that are compiled in a translation unit of their own (kernels_%s.hip), beside adlhip.hip instead of inside it." % name,
            "// X(signature): `extern template` in adlhip.hip, explicit instantiation in kernels_%s.hip.  An instantiation that is missing here" % name,
            "// is simply compiled with adlhip.hip; one that is listed but no longer used costs build time only."]
    open(os.path.join(ROOT, "oclradixsort_amd", "csrc", name + "_kernels.inc"), "w").write("\n".join(head + ["X(%s)" % s for s in sigs]) + "\n")
    print("%s: %d instantiations" % (name, len(sigs)))
