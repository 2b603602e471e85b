"""Copies the DATA the reference's own tests hold into tests/golden/ (runs only in
the build container where /root/reference exists):

* test/qapdata/esc16j.dat  -> esc16j.dat (QAPLIB instance, verbatim data file)
* the 64x64 integer matrix literal of test/numerical_issues.jl:1-66
  -> numerical_issues_P64.txt (whitespace-separated integers, one matrix row per line)
"""
import re, shutil, pathlib
ref = pathlib.Path("/root/reference/test")
out = pathlib.Path(__file__).parent
shutil.copy(ref / "qapdata" / "esc16j.dat", out / "esc16j.dat")
txt = (ref / "numerical_issues.jl").read_text()
body = txt[txt.index("["):txt.index("]")]
rows = [r.split() for r in body.strip("[").split(";")]
rows = [[int(t) for t in r] for r in rows if r]
assert len(rows) == 64 and all(len(r) == 64 for r in rows)
(out / "numerical_issues_P64.txt").write_text("\n".join(" ".join(map(str, r)) for r in rows) + "\n")
print("ok")
