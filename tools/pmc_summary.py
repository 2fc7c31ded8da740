"""Per-kernel summary of a rocprofv3 --pmc counter_collection.csv (sums over dispatches, kernel names shortened).
usage: python tools/pmc_summary.py <dir-or-csv> [min_share]   Dev / docs tool (profiles/r02_sq_counters.txt)."""
import collections, csv, glob, os, re, sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")
    if "at::native" in name[:60]:
        m = re.search(r"at::native::([A-Za-z_0-9]+)", name)
        return "torch:" + (m.group(1) if m else "op")
    return re.sub(r"\(.*$", "", name)[:70]


def main():
    src = sys.argv[1]
    files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    dur = collections.defaultdict(dict)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k].add(r["Dispatch_Id"])
            dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    counters = sorted({c for k in tot for c in tot[k]})
    print("counters:", " ".join(counters))
    rows = sorted(tot, key=lambda k: -sum(dur[k].values()))
    for k in rows:
        ms = sum(dur[k].values())
        if ms < float(sys.argv[2]) if len(sys.argv) > 2 else ms < 0.5:
            continue
        c = tot[k]
        line = f"{k:58s} launches {len(cnt[k]):5d}  time {ms:9.2f} ms"
        print(line)
        print("    " + "  ".join(f"{n}={c[n]:.4g}" for n in counters if n in c))
        d = []
        if c.get("SQ_INSTS_VALU") and c.get("SQ_INSTS_MFMA") is not None and "SQ_INSTS_MFMA" in c:
            mf = c["SQ_INSTS_MFMA"]
            if mf > 0:
                d.append(f"non-MFMA VALU per MFMA {(c['SQ_INSTS_VALU'] - mf) / mf:.2f}")
        if c.get("SQ_WAVE_CYCLES"):
            wc = c["SQ_WAVE_CYCLES"]
            for n in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
                if n in c:
                    d.append(f"{n}/WAVE_CYCLES {c[n] / wc:.3f}")
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("GRBM_GUI_ACTIVE"):
            # MfmaUtil (gfx94x formula of derived_counters.xml): MFMA-busy cycles summed over the SIMDs / (active cycles x CUs x 4);
            # rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back), hence the / 8
            act = c["GRBM_GUI_ACTIVE"] / 8.0
            d.append(f"MfmaUtil = MFMA_BUSY/(GUI_ACTIVE/8 x 256 CU x 4) {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (act * 256 * 4):.3f}")
            d.append(f"effective clock {act / (ms * 1e-3) / 1e9:.2f} GHz")
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("SQ_BUSY_CYCLES"):
            d.append(f"MFMA_BUSY/SQ_BUSY_CYCLES {c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_BUSY_CYCLES']:.3f}")
        hit, miss = c.get("TCC_HIT_sum"), c.get("TCC_MISS_sum")
        if hit is not None and miss is not None and hit + miss > 0:
            d.append(f"L2 hit rate {hit / (hit + miss):.3f}")
        if "FETCH_SIZE" in c:
            d.append(f"fetch {2 * c['FETCH_SIZE'] * 1024 / 1e9:.2f} GB total (2 x FETCH_SIZE KB, gfx950 correction)")
        if "WRITE_SIZE" in c:
            d.append(f"write {c['WRITE_SIZE'] * 1024 / 1e9:.2f} GB total")
        if d:
            print("    -> " + "; ".join(d))


if __name__ == "__main__":
    main()
