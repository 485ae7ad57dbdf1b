"""Lists (i) every `QS.<name>` attribute that the reference's Python files use and (ii) every method name of the reference's
_quisk method table (quisk.c:5164-5266), with the PyArg_ParseTuple format of its implementation where one is found.  Names and
argument formats are interface facts; this script only reads the reference as text.  Output: tests/golden/qs_names.json.
Run in the build container: python tests/golden/make_qs_names.py"""
import glob, json, os, re
REF = "/root/reference"
used = set()
for f in glob.glob(REF + "/**/*.py", recursive=True):
    try:
        txt = open(f, encoding="utf-8", errors="replace").read()
    except OSError:
        continue
    if not re.search(r"import _quisk as QS|import _quisk\b", txt):
        continue
    used.update(re.findall(r"\bQS\.([A-Za-z_][A-Za-z0-9_]*)", txt))
src = open(REF + "/quisk.c", encoding="utf-8", errors="replace").read()
table = src[src.index("static PyMethodDef QuiskMethods[]"):]
table = table[:table.index("};")]
methods = re.findall(r'\{\s*"([A-Za-z_0-9]+)"\s*,\s*(?:\(PyCFunction\))?\s*([A-Za-z_0-9]+)', table)
formats = {}
allsrc = "".join(open(f, encoding="utf-8", errors="replace").read() for f in glob.glob(REF + "/*.c"))
for name, fn in methods:
    m = re.search(r"PyObject\s*\*\s*" + re.escape(fn) + r"\s*\([^)]*\)\s*(?://[^\n]*)?\s*\{(.*?)\n\}", allsrc, re.S)
    fmt = None
    if m:
        k = re.search(r'PyArg_ParseTuple(?:AndKeywords)?\s*\(\s*args\s*,(?:\s*\w+\s*,)?\s*"([^"]*)"', m.group(1))
        if k:
            fmt = k.group(1)
    formats[name] = fmt
out = {"used_by_python": sorted(used), "method_table": [n for n, _ in methods], "formats": formats}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "qs_names.json"), "w"), indent=1, sort_keys=True)
print(len(used), "names used,", len(methods), "in the table")
