"""Random vid configurations of ONE extra field on the reference's t{0,1,2}_asa fixtures: an INFO field (length A / R / G / 1 / 2 / 3 /
VAR, int or float, one of the combine operations the vid mapper accepts for it) or a FORMAT field (same lengths and types), present in
most records with random values and some missing elements.  inputs(seed, cls, tmp) -> (cells, query, description); the callers compare
the oracle with the kernel bodies under g++ (CPU suite) and with the device (GPU suite).
usage: python tests/tools/field_fuzz.py INFO|FORMAT first_seed last_seed   (oracle against kernel bodies, prints the mismatches)"""
import gzip, json, os, random, sys
_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
import helpers
import vcf2cells

SAMPLES = [("t0_asa", "HG00141"), ("t1_asa", "HG01958"), ("t2_asa", "HG01530")]


def inputs(seed, cls, tmp):
    rnd = random.Random(seed * 2 + (cls == "FORMAT"))
    length = rnd.choice(["A", "R", "G", 1, 2, 3, "VAR"])
    typ = rnd.choice(["int", "float"])
    desc = {"vcf_field_class": [cls], "type": typ, "length": length}
    op = None
    if cls == "INFO":
        op = rnd.choice(["sum", "mean", "median", "element_wise_sum"] + (["concatenate"] if length == "VAR" else []))
        desc["VCF_field_combine_operation"] = op
    src_vid = json.load(open(os.path.join(helpers.GOLDEN, "inputs", "vid_all_asa.json")))
    fields = {}
    for k, v in src_vid["fields"].items():
        fields[k] = v
        if k == "MQ0":
            fields["XF"] = desc
    vp = os.path.join(tmp, "vid.json")
    with open(vp, "w") as f:
        json.dump(dict(src_vid, fields=fields), f)
    callsets = {"callsets": {}}
    for row, (name, sample) in enumerate(SAMPLES):
        with gzip.open(os.path.join(helpers.GOLDEN, "inputs", "vcfs", name + ".vcf.gz"), "rt") as f:
            lines = f.read().splitlines()
        out = []
        for l in lines:
            if l.startswith("#CHROM"):
                out.append('##%s=<ID=XF,Number=%s,Type=%s,Description="x">' % (cls, "." if length == "VAR" else str(length), "Integer" if typ == "int" else "Float"))
            if not l.startswith("#"):
                c = l.split("\t")
                if rnd.random() < 0.85:
                    nal = 1 + len(c[4].split(","))
                    n = {"A": nal - 1, "R": nal, "G": nal * (nal + 1) // 2, "VAR": rnd.randint(1, 5)}.get(length, length)
                    vals = []
                    for _ in range(n):
                        if rnd.random() < 0.1:
                            vals.append(".")
                        elif typ == "int":
                            vals.append(str(rnd.randint(-50, 300)))
                        else:
                            vals.append("%.3f" % rnd.uniform(-5, 50))
                    if cls == "INFO":
                        extra = "XF=" + ",".join(vals)
                        c[7] = extra if c[7] in (".", "") else c[7] + ";" + extra
                    else:
                        c[8] += ":XF"
                        c[9] += ":" + ",".join(vals)
                l = "\t".join(c)
            out.append(l)
        p = os.path.join(tmp, name + ".vcf.gz")
        with gzip.open(p, "wt") as f:
            f.write("\n".join(out) + "\n")
        callsets["callsets"][sample] = {"row_idx": row, "idx_in_file": 0, "filename": p}
    cp = os.path.join(tmp, "callsets.json")
    with open(cp, "w") as f:
        json.dump(callsets, f)
    cells = b"".join(c[3] for c in vcf2cells.build_cells(cp, vp, lambda fn: fn))
    q = {"vid_mapping_file": vp, "callset_mapping_file": cp, "vcf_header_filename": os.path.join(helpers.GOLDEN, "inputs", "template_vcf_header.vcf"),
         "reference_genome": os.path.join(helpers.GOLDEN, "inputs", "chr1_10MB.fasta.gz"), "query_column_ranges": [[[0, helpers.INT64_MAX - 1]]]}
    return cells, q, "%s length %s %s %s" % (cls, length, typ, op or "")


def inputs_on_fixture(seed, callsets, vid_name, tmp):
    """the same on ANY of the reference's callset fixtures (haploid / triploid calls, spanning deletions, overlapping intervals, several
    samples per file): one extra INFO or FORMAT field per record and sample, G-length vectors sized by the sample's ploidy; INFO fields
    also with move_to_FORMAT.  -> (cells, vid path, callsets path, description)"""
    import math
    rnd = random.Random(seed)
    cls = rnd.choice(["INFO", "FORMAT"])
    length = rnd.choice(["A", "R", "G", 1, 2, "VAR"])
    typ = rnd.choice(["int", "float"])
    desc = {"vcf_field_class": [cls], "type": typ, "length": length}
    if cls == "INFO":
        desc["VCF_field_combine_operation"] = rnd.choice(["sum", "mean", "median", "element_wise_sum", "move_to_FORMAT"] + (["concatenate"] if length == "VAR" else []))
    with open(os.path.join(helpers.GOLDEN, "inputs", vid_name)) as f:
        vid = json.load(f)
    vid["fields"]["XF"] = desc
    vp = os.path.join(tmp, "vid.json")
    with open(vp, "w") as f:
        json.dump(vid, f)
    with open(os.path.join(helpers.GOLDEN, "inputs", "callsets", callsets)) as f:
        cs = json.load(f)
    out_cs = {"callsets": {}}
    files = {}
    for sample, info in sorted(cs["callsets"].items(), key=lambda kv: kv[1]["row_idx"]):
        fn = info["filename"]
        if fn not in files:
            with gzip.open(os.path.join(helpers.GOLDEN, fn), "rt") as f:
                lines = f.read().splitlines()
            out = []
            for l in lines:
                if l.startswith("#CHROM"):
                    out.append('##%s=<ID=XF,Number=%s,Type=%s,Description="x">' % (cls, "." if length == "VAR" else str(length), "Integer" if typ == "int" else "Float"))
                if not l.startswith("#"):
                    c = l.split("\t")
                    nal = 1 + len(c[4].split(","))

                    def vals(ploidy):
                        n = {"A": nal - 1, "R": nal, "G": math.comb(ploidy + nal - 1, nal - 1), "VAR": rnd.randint(1, 4)}.get(length, length)
                        return ",".join("." if rnd.random() < 0.1 else (str(rnd.randint(-20, 200)) if typ == "int" else "%.2f" % rnd.uniform(-3, 40)) for _ in range(n))
                    fmt = c[8].split(":")
                    gts = [s.split(":")[fmt.index("GT")].replace("|", "/").split("/") if "GT" in fmt else ["0", "0"] for s in c[9:]]
                    if rnd.random() < 0.85:
                        if cls == "INFO":
                            e = "XF=" + vals(len(gts[0]))
                            c[7] = e if c[7] in (".", "") else c[7] + ";" + e
                        else:
                            c[8] += ":XF"
                            for i in range(len(c) - 9):
                                c[9 + i] += ":" + vals(len(gts[i]))
                    l = "\t".join(c)
                out.append(l)
            p = os.path.join(tmp, "f%d.vcf.gz" % len(files))
            with gzip.open(p, "wt") as f:
                f.write("\n".join(out) + "\n")
            files[fn] = p
        out_cs["callsets"][sample] = dict(info, filename=files[fn])
    cp = os.path.join(tmp, "callsets.json")
    with open(cp, "w") as f:
        json.dump(out_cs, f)
    cells = b"".join(c[3] for c in vcf2cells.build_cells(cp, vp, lambda fn: fn))
    return cells, vp, cp, "%s length %s %s %s" % (cls, length, typ, desc.get("VCF_field_combine_operation", ""))


if __name__ == "__main__":
    import tempfile
    bad = 0
    for seed in range(int(sys.argv[2]), int(sys.argv[3])):
        cells, q, what = inputs(seed, sys.argv[1], tempfile.mkdtemp())
        txt, _, _ = helpers.oracle_run(q, cells, with_header=False)
        got, err = helpers.hostsim_run(q, cells, with_header=False)
        if err or got != txt:
            bad += 1
            print("seed", seed, what, "MISMATCH", err)
    print("field_fuzz:", int(sys.argv[3]) - int(sys.argv[2]), "cases,", bad, "mismatches")
