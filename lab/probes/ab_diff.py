"""Compare ab_lib.py outputs: python ab_diff.py A.json B.json [A2.json B2.json ...] (A = first lib, B = second)."""
import json
import sys
runs = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sys.argv[1:]]
A, B = runs[0::2], runs[1::2]
fa = min(min(r["forward_ms"]) for r in A); fb = min(min(r["forward_ms"]) for r in B)
print(f"forward: {A[0]['lib']} {fa:.3f} ms   {B[0]['lib']} {fb:.3f} ms   ({(fb / fa - 1) * 100:+.1f}%)")
n = len(A[0]["launches"])
ta = [min(r["launches"][i][1] for r in A) for i in range(n)]
tb = [min(r["launches"][i][1] for r in B) for i in range(n)]
print(f"sum of launches: {sum(ta):.1f} us vs {sum(tb):.1f} us")
rows = sorted(range(n), key=lambda i: tb[i] - ta[i])
for i in rows[:12] + rows[-12:]:
    print(f"{i:3d} {ta[i]:8.1f} -> {tb[i]:8.1f} us ({tb[i] - ta[i]:+6.1f})  {A[0]['launches'][i][0]}")
