import sys; sys.path.insert(0, "/root/repo")
from tests import smoke_impl as S
import io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    res = S.compare_step(4, 64, 3, True, verbose=True)
lines = [l for l in buf.getvalue().splitlines() if "rel_pos" in l or "cos" in l]
import re
rows = []
for l in buf.getvalue().splitlines():
    m = re.match(r"(\S+)\s+cos ([0-9.]+)\s+\|ref\| (\S+) \|got\| (\S+)", l)
    if m: rows.append((m.group(1), float(m.group(2)), float(m.group(3)), float(m.group(4))))
rows.sort(key=lambda r: abs(r[3] / r[2] - 1), reverse=True)
tot = sum(r[2] ** 2 for r in rows) ** 0.5
for r in rows[:12]:
    print("%-40s cos %.4f ratio %.4f  |ref|/|total| %.2e" % (r[0], r[1], r[3] / r[2], r[2] / tot))
print({k: v for k, v in res.items() if "grad" in k})
