"""Condenses one rocprofv3 --pmc database into a JSON of mean counter values per kernel:
python tools/sq_counters.py <results.db> <out.json> <command text> <kernel name part> ..."""
import json, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
names = sys.argv[4:]
out = {}
for kname, cname, mean, cnt in db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                          "group by kernel_name, counter_name"):
    if not names or any(k in kname for k in names):
        out.setdefault(kname, {})[cname] = mean
        out[kname]["launches"] = cnt
json.dump({"command": sys.argv[3], "mean_per_launch": out}, open(sys.argv[2], "w"), indent=1)
for k, v in out.items():
    print(k[:90], {a: round(b) for a, b in v.items()})
