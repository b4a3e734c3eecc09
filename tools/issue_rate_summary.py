"""profiles/rNN_issue_rate.txt (the table tools/issue_rate.hip prints on the GPU box) -> profiles/rNN_issue_rate.json, the record
bench.py's `roofline.issue` reads: ns a SIMD spends per wave-instruction it retires, by instruction mix and waves per SIMD.
usage: issue_rate_summary.py profiles/r05_issue_rate.txt > profiles/r05_issue_rate.json"""
import json
import re
import sys

VPS = {"valu independent": None, "same wave 1 valu : 1 salu": 1.0, "same wave 2 valu : 1 salu": 2.0, "same wave 4 valu : 1 salu": 4.0}
mixes = {}
clk = []
for ln in open(sys.argv[1]):
    if ln.startswith("#") or ln.startswith("mix"):
        continue
    m = re.match(r"(.+?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)(?:\s+([\d.]+)\s+([\d.]+))?\s*$", ln)
    if not m:
        continue
    name, W = m.group(1).strip(), int(m.group(2))
    e = mixes.setdefault(name, dict(name=name, valu_per_salu=None, ns_per_inst_per_simd={}, cyc_per_inst_per_wave={}))
    e["ns_per_inst_per_simd"][str(W)] = float(m.group(7))
    e["cyc_per_inst_per_wave"][str(W)] = float(m.group(4))
    for key, v in VPS.items():
        if name.startswith(key):
            e["valu_per_salu"] = v if v is not None else 64.0  # (pure VALU: stands for "hardly any scalar work")
    if m.group(8) and W == 1:
        clk.append(float(m.group(8)))
out = dict(source=sys.argv[1], clock_ghz=round(sum(clk) / len(clk) / 1e3, 3) if clk else 2.4,
           mixes=list(mixes.values()),
           note="ns/inst/SIMD = wall time of a launch that fills every SIMD of the chip with W waves of the mix / instructions per wave / W; "
                "bench.py prices a kernel's wave-instructions with the row whose VALU : SALU ratio is closest to the kernel's")
print(json.dumps(out, indent=1))
