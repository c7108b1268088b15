"""profiles/rNN_marcher_traffic.json from a tools/pmc_run.sh summary (FETCH_SIZE / WRITE_SIZE passes of `bench.py`): HBM-side bytes per
launch of the three marcher kernels, stamped with the kernel source fingerprint of THIS tree (bench.py reports a mismatch as STALE).
usage: python tools/make_traffic_json.py <pmc_summary.md> <out.json> <commit>"""
import json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
md, out, commit = sys.argv[1:4]
comp, cur = {}, None
for line in open(md):
    m = re.match(r'## (?:void )?(.+)$', line)
    if m:
        full = m.group(1).strip()
        cur = full.split('<')[0]
        if cur == 'k4_geom3_kernel' and 'true' in full:          # the sample-counting instantiation (bench's counter frames): not the product launch
            cur = None
        continue
    m = re.match(r'\| (FETCH_SIZE|WRITE_SIZE) \| ([0-9.e+]+) \|', line)
    if m and cur:
        comp.setdefault(cur, {})[m.group(1) + '_KiB'] = float(m.group(2))
    # round 4: the instruction mix and the L1 -> L2 requests of the same launches (other passes of the same command), for the second roof
    m = re.match(r'\| (SQ_INSTS_VALU|SQ_INSTS_MFMA|SQ_INSTS_SALU|SQ_INSTS_LDS|SQ_INSTS_VMEM_RD|SQ_INSTS_VMEM_WR|SQ_VALU_MFMA_BUSY_CYCLES|GRBM_GUI_ACTIVE|'
                 r'TCP_TCC_READ_REQ_sum|TCP_TCC_WRITE_REQ_sum|TCC_HIT_sum|TCC_MISS_sum) \| ([0-9.e+]+) \|', line)
    if m and cur:
        comp.setdefault(cur, {})[m.group(1)] = float(m.group(2))
want = ('k4_geom3_kernel', 'k4_order_kernel', 'k4_shade_kernel')
sel = {k: v for k, v in comp.items() if k in want}
assert set(sel) == set(want), (list(comp), 'marcher kernels missing from the summary')
raw = sum(v.get('FETCH_SIZE_KiB', 0) + v.get('WRITE_SIZE_KiB', 0) for v in sel.values()) * 1024
# MI355X_MICROARCH.md "HBM": on gfx950 FETCH_SIZE reports half of the bytes of 16-B-per-lane reads.  The shading kernel's reads are
# 16-byte k0 gathers (24 per sample) -> doubled; the geometry kernel reads 4-byte density corners (calibrated exact on k_repack_k0's
# 4-byte stream in round 2) -> as reported.  WRITE_SIZE as reported (calibrated exact on k_rays_of_view in round 2).
corr = raw + sel['k4_shade_kernel'].get('FETCH_SIZE_KiB', 0) * 1024
json.dump({'fabric_bytes_per_launch': int(corr), 'fabric_bytes_per_launch_as_reported': int(raw),
           'correction': 'k4_shade_kernel FETCH_SIZE x 2 (16-byte gathers report half on gfx950, MI355X_MICROARCH.md HBM section); everything else as reported',
           'components': sel, 'commit': commit, 'kernel_source_sha1': bench.marcher_source_sha1(),
           'source': f'{os.path.basename(md)}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/pmc_run.sh), mean per launch, LLFF synthetic frame'},
          open(out, 'w'), indent=1)
print(open(out).read())
