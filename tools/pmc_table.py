"""Per-kernel table from rocprofv3 --pmc passes (tools/pmc_table.sh):  python tools/pmc_table.py <name substring> <dirs...>
Raw counter averages per dispatch and what they say about the binding resource.  Units (MI355X_MICROARCH.md): the SQ
cycle counters count QUAD-cycles summed over waves (or CUs); GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import collections
import csv
import glob
import sys

pat, dirs = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if pat not in k:
                continue
            k = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
CUS, SIMDS = 256, 1024
for k, d in sorted(acc.items()):
    m = {c: sum(v[len(v) // 3:]) / len(v[len(v) // 3:]) for c, v in d.items()}    # (the first third: warm-up dispatches)
    print(k, f"   ({len(next(iter(d.values())))} dispatches per pass)")
    for c in sorted(m):
        print(f"   {c:26s} {m[c]:16.0f}")
    g = lambda c: m.get(c, float("nan"))
    cyc = g("GRBM_GUI_ACTIVE") / 8.0                      # shader cycles the dispatch was active (per XCD)
    simd_quads = cyc * SIMDS / 4.0                        # quad-cycles available on all SIMDs
    print("   -- derived")
    print(f"   active cycles per XCD              {cyc:12.0f}   (~{cyc / 2.1e3:.1f} us at 2.1 GHz)")
    print(f"   waves per SIMD, time average        {g('SQ_WAVE_CYCLES') / simd_quads:12.2f}   (SQ_WAVE_CYCLES / SIMD quad-cycles; 8 = full)")
    print(f"   VALU busy, fraction of SIMD time   {g('SQ_ACTIVE_INST_VALU') / simd_quads:12.3f}   (peak 1.0: one VALU instruction in flight per SIMD)")
    print(f"   scalar issue, fraction of SIMD time {g('SQ_ACTIVE_INST_SCA') / simd_quads:11.3f}")
    print(f"   CU busy, fraction of the dispatch  {g('SQ_BUSY_CU_CYCLES') / (cyc * CUS):12.3f}   (SQ_BUSY_CU_CYCLES counts cycles, not quad-cycles)")
    wc = g("SQ_WAVE_CYCLES")
    print(f"   of a wave's life: waiting (waitcnt / barrier) {g('SQ_WAIT_ANY') / wc:6.3f}, issue stalls {g('SQ_WAIT_INST_ANY') / wc:6.3f}, "
          f"issuing {g('SQ_ACTIVE_INST_ANY') / wc:6.3f} (VALU {g('SQ_ACTIVE_INST_VALU') / wc:5.3f}, scalar {g('SQ_ACTIVE_INST_SCA') / wc:5.3f}, "
          f"VMEM {g('SQ_ACTIVE_INST_VMEM') / wc:5.3f}, LDS {g('SQ_ACTIVE_INST_LDS') / wc:5.3f})")
    w = g("SQ_WAVES")
    print(f"   per wavefront: {g('SQ_INSTS_VALU') / w:9.0f} VALU, {g('SQ_INSTS_SALU') / w:8.0f} SALU, {g('SQ_INSTS_SMEM') / w:6.0f} SMEM, "
          f"{g('SQ_INSTS_VMEM') / w:6.0f} VMEM, {g('SQ_INSTS_LDS') / w:6.0f} LDS instructions; {w:.0f} wavefronts")
    print(f"   VALU lane utilisation              {g('SQ_THREAD_CYCLES_VALU') / (64.0 * g('SQ_ACTIVE_INST_VALU')):12.3f}   (active lanes per VALU quad-cycle / 64)")
    print(f"   vector-memory instructions in flight per busy CU  {4.0 * g('SQ_INST_LEVEL_VMEM') / g('SQ_BUSY_CU_CYCLES'):8.2f}   (SQ_INST_LEVEL_VMEM x 4 / SQ_BUSY_CU_CYCLES)")
    print(f"   cycles per VALU instruction        {4.0 * g('SQ_ACTIVE_INST_VALU') / g('SQ_INSTS_VALU'):12.2f}   (4 = a wave64 fp32 instruction's issue slot; packed / transcendental: 8)")
