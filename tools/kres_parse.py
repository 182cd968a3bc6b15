import sys, re, subprocess
rows, cur = [], None
for line in sys.stdin:
    if re.match(r'\s+- \.agpr_count:', line):
        if cur: rows.append(cur)
        cur = {}
    m = re.match(r'\s+-?\s*\.(\w+):\s+(.*)', line)
    if m and cur is not None: cur[m.group(1)] = m.group(2).strip()
if cur: rows.append(cur)
names = subprocess.run(["c++filt"], input="\n".join(r.get("name","?") for r in rows), capture_output=True, text=True).stdout.split("\n")
for r, n in zip(rows, names):
    print(f"{r.get('vgpr_count','?'):>4} vgpr {r.get('agpr_count','?'):>3} agpr {r.get('sgpr_count','?'):>3} sgpr spill {r.get('vgpr_spill_count','?')} lds {r.get('group_segment_fixed_size','?'):>6}  {n[:150]}")
