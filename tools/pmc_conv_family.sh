# Dev tool: HBM-side traffic of the sparse-convolution FAMILY of one eval forward (every k_conv_* launch + k_concat2_idn):
# two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass) over tools/prof_forward.py, summed per forward.
# Writes gpurun_out/pmc_family/pmc_conv_family_latest.json (copy into profiles/): bench.py quotes it as roofline.traffic
# while the hashes of the kernel sources match.
export GPU_MAX_HW_QUEUES=16  # (in this shell: the profiler brings the GPU up before python starts)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/pmc_family; rm -rf $O; mkdir -p $O
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/g$i -- python3 $R/tools/prof_forward.py 4 > /dev/null 2>&1
  echo "group $i ($grp) rc=$?"
done
python3 - <<PY
import csv, glob, collections, json, hashlib
fam = ('k_conv_g16p', 'k_conv_lw', 'k_conv_os', 'k_conv_flat', 'k_conv_pair', 'k_conv_g16<', 'k_concat2_idn')
tot = collections.defaultdict(float); nfwd = {}; launches = {}
per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('$O/g*/*/*counter_collection.csv'):
    n = 0; nl = 0
    for r in csv.DictReader(open(f)):
        name, c = r['Kernel_Name'], r['Counter_Name']
        if 'k_voxelize_fp' in name: n += 1
        if any(k in name for k in fam):
            tot[c] += float(r['Counter_Value']); nl += 1
            per_kernel[name.split('(')[0].replace('void ', '')][c] += float(r['Counter_Value'])
        ctr = c
    nfwd[ctr] = n; launches[ctr] = nl
print(dict(tot), nfwd, launches)
fw = min(nfwd.values())
fetch, write = tot['FETCH_SIZE'] / nfwd['FETCH_SIZE'], tot['WRITE_SIZE'] / nfwd['WRITE_SIZE']
cor = (2 * fetch + write) * 1024
sha = hashlib.sha256(b''.join(open('$R/geoformer_amd/csrc/' + s, 'rb').read() for s in ('spconv_conv.hip', 'spconv_lw.hip', 'unet_exec.hip'))).hexdigest()
json.dump({"bytes_per_forward": int(cor), "fetch_kib_per_forward": fetch, "write_kib_per_forward": write,
           "bytes_raw_per_forward": int((fetch + write) * 1024), "forwards": fw, "launches_per_forward": launches['FETCH_SIZE'] / nfwd['FETCH_SIZE'],
           "per_kernel_kib_per_forward": {k: {c: v / nfwd[c] for c, v in d.items()} for k, d in per_kernel.items()},
           "kernel_source_sha256": sha,
           "source": "tools/pmc_conv_family.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/prof_forward.py "
                     "(S150k benchmark scene), every k_conv_* launch and k_concat2_idn of a forward summed; FETCH_SIZE x2 per "
                     "MI355X_MICROARCH.md (gfx950), KiB units; the hash covers spconv_conv.hip + spconv_lw.hip + unet_exec.hip"},
          open('$O/pmc_conv_family_latest.json', 'w'), indent=1)
print(open('$O/pmc_conv_family_latest.json').read()[:1500])
PY
