import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value",round(d["value"]),"steady",d.get("value_steady"),d.get("value_windows"))
for c in d.get("configs",[]): print(c)
print([ (m["sectors_per_gpu"], round(m["scans_per_sec"]), round(m["x_single_sector"],2), round(m["roofline"]["frac"],4), round(m["roofline"]["grow_us_per_scan_all_groups"],1), round(m["roofline"]["grow_us_per_group_launch"],1)) for m in d["multi_sector_all"]])
print("api", d["api_scans_per_sec"], "roofline", d["roofline"]["frac"], d["roofline"]["traffic"], "cpu", d["cpu_baseline"]["value"] if d["cpu_baseline"] else None)
