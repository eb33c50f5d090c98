#!/usr/bin/env python
"""Turn a rocprofv3 `--kernel-trace --stats --output-format csv` kernel_stats.csv into the markdown table kept in profiles/."""
import csv, sys
path, title = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ''
rows = list(csv.DictReader(open(path)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(title)
print('| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|')
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:28]:
    nm = r['Name']
    if nm.startswith('void at::') or 'rocclr' in nm:
        nm = nm[:60] + '… (torch init/plumbing)'
    print(f"| `{nm[:110]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {100 * float(r['TotalDurationNs']) / tot:.2f} |")
