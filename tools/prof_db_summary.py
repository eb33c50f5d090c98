#!/usr/bin/env python
"""Markdown table of the `top_kernels` view of a rocprofv3 (ROCm 7.2) results .db (`--kernel-trace --stats`), as kept in profiles/."""
import sqlite3, sys
db, title = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ''
rows = list(sqlite3.connect(db).execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
print(title)
print('| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|')
for nm, calls, tot, avg, pct in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 26]:
    if nm.startswith('void at::') or 'rocclr' in nm:
        nm = nm[:70] + '... (torch plumbing: RNG / fill / copy)'
    print(f"| `{nm[:120]}` | {calls} | {tot / 1e3:.2f} | {avg:.1f} | {pct:.2f} |")
