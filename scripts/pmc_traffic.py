"""HBM bytes per launch from the request-size counters of a `gpu_profile_round.sh` PMC summary (profiles/rNN_pmc_traffic.txt):
   reads  = RDREQ_32B x 32 + RDREQ_64B x 64 + RDREQ_128B x 128,  writes = WRREQ_64B x 64 + (WRREQ - WRREQ_64B) x 32
(MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE derive from these fabric-side request counters and are
uncalibrated on gfx950; the request sizes are exact).  Writes profiles/pmc_traffic.json, which bench.py quotes as
roofline.traffic with its source.   usage: pmc_traffic.py SUMMARY.txt CHUNK "source text" [out.json]
   round 5:  pmc_traffic.py --cfg34 SUMMARY_CFG34.txt "source text" [out.json]  merges the config 3 / 4 / -e45 kernels (sections
   "#### codec" of a gpu_pmc.sh log per coder; two-pass encoders: both passes summed) into the existing file"""
import json
import re
import sys

def bytes_of(v):
    rd = v.get("TCC_EA0_RDREQ_32B_sum", 0) * 32 + v.get("TCC_EA0_RDREQ_64B_sum", 0) * 64 + v.get("TCC_EA0_RDREQ_128B_sum", 0) * 128
    wr = v.get("TCC_EA0_WRREQ_64B_sum", 0) * 64 + (v.get("TCC_EA0_WRREQ_sum", 0) - v.get("TCC_EA0_WRREQ_64B_sum", 0)) * 32
    return rd, wr


if sys.argv[1] == "--cfg34":
    txt, source = sys.argv[2], sys.argv[3]
    out = sys.argv[4] if len(sys.argv) > 4 else "profiles/pmc_traffic.json"
    KERN = {"rccdf": (1536, ["trc_rca_enc_mc_kernel"], ["trc_rca_dec_kernel"]),
            "anscdf": (1536, ["trc_ansa_model2_kernel", "trc_ansa_codeq_kernel"], ["trc_ansa_dec_kernel"]),
            "rcs": (1536, ["trc_rcb_enc_mc_kernel"], ["trc_rcb_dec_kernel"]),
            "rccdfs2": (1024, ["trc_rcs2p_enc_kernel"], ["trc_rcs2p_dec_kernel"]),
            "anscdf1": (4096, ["trc_o1_sort_kernel", "trc_o1_walk_kernel", "trc_o1_place_kernel", "trc_ansa_codeq_kernel"], ["trc_o1_dec_rows"])}
    sect, cur, vals = None, None, {}
    for line in open(txt):
        if line.startswith("####"):
            sect = line.split()[1]; cur = None
            continue
        m = re.match(r"\s+(\S+)\s+mean\s+([\d.]+)", line)
        if m and cur and sect:
            vals.setdefault(sect, {}).setdefault(cur, {})[m.group(1)] = float(m.group(2))
        elif line.strip() and not line.startswith("==") and not line.startswith(" "):
            cur = line.strip()
    res = json.load(open(out))
    res["source_cfg34"] = source
    for codec, (chunk, enc, dec) in KERN.items():
        for tag, names in (("enc", enc), ("dec", dec)):
            rd = wr = 0
            for kn in names:
                v = next((vals.get(codec, {})[n] for n in vals.get(codec, {}) if kn in n), None)
                if v:
                    a, b = bytes_of(v); rd += a; wr += b
            if rd + wr:
                res["%s_%s_chunk%d" % (codec, tag, chunk)] = int(rd + wr)
                res["%s_%s_chunk%d_read" % (codec, tag, chunk)] = int(rd)
                res["%s_%s_chunk%d_written" % (codec, tag, chunk)] = int(wr)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))
    sys.exit(0)

txt, chunk, source = sys.argv[1], int(sys.argv[2]), sys.argv[3]
out = sys.argv[4] if len(sys.argv) > 4 else "profiles/pmc_traffic.json"
cur, vals = None, {}
for line in open(txt):
    m = re.match(r"\s+(\S+)\s+mean\s+([\d.]+)", line)
    if m and cur:
        vals.setdefault(cur, {})[m.group(1)] = float(m.group(2))
    elif line.strip() and not line.startswith("==") and not line.startswith(" "):
        cur = line.strip()
keys = {"anscdf4s_enc": "trc_ans4s_enc_kernel", "anscdf4s_dec": "trc_ans4s_dec_kernel", "gather": "trc_gather_kernel"}
res = {"source": source}
for k, kern in keys.items():
    v = next((vals[n] for n in vals if kern in n), None)
    if not v:
        continue
    rd = v.get("TCC_EA0_RDREQ_32B_sum", 0) * 32 + v.get("TCC_EA0_RDREQ_64B_sum", 0) * 64 + v.get("TCC_EA0_RDREQ_128B_sum", 0) * 128
    wr = v.get("TCC_EA0_WRREQ_64B_sum", 0) * 64 + (v.get("TCC_EA0_WRREQ_sum", 0) - v.get("TCC_EA0_WRREQ_64B_sum", 0)) * 32
    res["%s_chunk%d" % (k, chunk)] = int(rd + wr)
    res["%s_chunk%d_read" % (k, chunk)] = int(rd)
    res["%s_chunk%d_written" % (k, chunk)] = int(wr)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
