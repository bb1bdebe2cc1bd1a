#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2) rocpd sqlite result: per-kernel calls / total / average / share,
i.e. the `--stats` table, written as markdown.  usage: rocpd_summary.py results.db [out.md] [title]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
  name = re.sub(r'\(anonymous namespace\)::', '', name)
  name = re.sub(r'\((?:[^()]|\([^()]*\))*\)\s*(\[clone .*\])?$', '', name)
  return name.replace('void ', '')[:110]


def main():
  db = sys.argv[1]
  out = sys.argv[2] if len(sys.argv) > 2 else None
  title = sys.argv[3] if len(sys.argv) > 3 else db
  c = sqlite3.connect(db)
  rows = c.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration) '
                   'from kernels group by name order by sum(duration) desc').fetchall()
  total = sum(r[2] for r in rows)
  span = c.execute('select min(start), max(end) from kernels').fetchone()
  lines = ['# %s' % title, '',
           'rocprofv3 --kernel-trace --stats; %d kernel dispatches, %.3f ms of kernel time, %.3f ms first-start to last-end.'
           % (sum(r[1] for r in rows), total / 1e6, (span[1] - span[0]) / 1e6), '',
           '| kernel | calls | total ms | avg us | min us | max us | % |', '|---|---:|---:|---:|---:|---:|---:|']
  for n, cnt, tot, avg, mn, mx in rows:
    lines.append('| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.2f |' % (short(n), cnt, tot / 1e6, avg / 1e3, mn / 1e3,
                                                                   mx / 1e3, 100.0 * tot / total))
  txt = '\n'.join(lines) + '\n'
  if out:
    open(out, 'w').write(txt)
  else:
    sys.stdout.write(txt)


if __name__ == '__main__':
  main()
