"""A second, independent restatement of the Accounter path in pure Python — written from the reference text
(pkg/flow/account.go:58-124, pkg/model/flow_content.go:28-61, bpf/flows.c:76-143), sharing no code with
oracle/nfagg_oracle.c — run against the C oracle on random streams. The parts of the path the reference's own tests do
not pin (order-dependent fields, wrap-around, the kernel dedup merge applied to whole records) thus have two
restatements that must agree bit for bit."""
import numpy as np
import pytest

U64, U32 = (1 << 64) - 1, (1 << 32) - 1
MAX_OBSERVED, DIR_BOTH, SERVER_HELLO, SSL_MISMATCH = 6, 3, 0x02, 0x01


def accumulate_base(p: dict, o: dict):
    """flow_content.go:28-61."""
    if p["start"] == 0 or (p["start"] > o["start"] and o["start"] != 0):
        p["start"] = o["start"]
    if p["end"] == 0 or p["end"] < o["end"]:
        p["end"] = o["end"]
    p["bytes"] = (p["bytes"] + o["bytes"]) & U64
    p["packets"] = (p["packets"] + o["packets"]) & U32
    p["flags"] |= o["flags"]
    if o["eth_protocol"] != 0:
        p["eth_protocol"] = o["eth_protocol"]
    if not any(p["src_mac"]):
        p["src_mac"] = o["src_mac"]
    if not any(p["dst_mac"]):
        p["dst_mac"] = o["dst_mac"]
    if o["dscp"] != 0:
        p["dscp"] = o["dscp"]
    if o["sampling"] != 0:
        p["sampling"] = o["sampling"]


def update_existing_flow(a: dict, o: dict):
    """bpf/flows.c:98-143 with a whole record in the place of one packet (packets += record.packets, tls_* from the
    record's fields), add_observed_intf :76-96."""
    ifx = o["if_index_first_seen"]
    if a["if_index_first_seen"] == ifx:
        a["packets"] = (a["packets"] + o["packets"]) & U32
        a["bytes"] = (a["bytes"] + o["bytes"]) & U64
        a["end"] = o["end"]
        a["flags"] |= o["flags"]
        a["dscp"], a["sampling"] = o["dscp"], o["sampling"]
        if o["ssl_version"] > 0 and a["ssl_version"] != o["ssl_version"]:
            if a["ssl_version"] == 0:
                a["ssl_version"] = o["ssl_version"]
            else:
                a["misc_flags"] |= SSL_MISMATCH
        if o["tls_cipher_suite"] > 0 and o["tls_types"] == SERVER_HELLO:
            a["tls_cipher_suite"] = o["tls_cipher_suite"]
        if o["tls_key_share"] > 0 and o["tls_types"] == SERVER_HELLO:
            a["tls_key_share"] = o["tls_key_share"]
        a["tls_types"] |= o["tls_types"]
    elif ifx != 0:
        a["end"] = o["end"]
        a["flags"] |= o["flags"]
        if a["nb_observed_intf"] >= MAX_OBSERVED:
            return
        for i in range(a["nb_observed_intf"]):
            if a["observed_intf"][i] == ifx:
                if a["observed_direction"][i] != o["direction_first_seen"] and a["observed_direction"][i] != DIR_BOTH:
                    a["observed_direction"][i] = DIR_BOTH
                return
        a["observed_intf"][a["nb_observed_intf"]] = ifx
        a["observed_direction"][a["nb_observed_intf"]] = o["direction_first_seen"]
        a["nb_observed_intf"] += 1


def to_dict(m) -> dict:
    d = {k: (m[k].tolist() if m[k].ndim else int(m[k])) for k in m.dtype.names}
    d["pad2"], d["pad4"] = [0, 0], [0, 0, 0, 0]                  # blank fields never reach Go (binary.Read skips them)
    return d


def account(recs, max_entries, dedup):
    """account.go:58-100: returns the evicted batches [(reason, {key bytes: metrics dict})]."""
    entries, out = {}, []
    for r in recs:
        key = r["id"].tobytes()[:39]                             # byte 39 is a blank field of BpfFlowId
        o = to_dict(r["metrics"])
        if key in entries:
            (update_existing_flow if dedup else accumulate_base)(entries[key], o)
        else:
            if len(entries) >= max_entries:
                out.append(("full", entries)); entries = {}
            entries[key] = o
    out.append(("closing", entries))
    return out


@pytest.mark.parametrize("dedup", [False, True])
@pytest.mark.parametrize("seed,n,keys,max_entries", [(1, 6000, 40, 1 << 20), (2, 8000, 700, 300), (3, 3000, 3, 2), (4, 5000, 200, 1 << 20)])
def test_c_oracle_agrees_with_the_python_restatement(O, dedup, seed, n, keys, max_entries):
    th = O.zipf_thresholds(keys, 1.1) if keys > 3 else None
    recs = O.gen_stream(n, seed=seed, n_keys=keys, thresholds=th, variant=2 if dedup else 1, hot_permille=300 if seed == 4 else 0)
    if seed % 2 == 0:                                            # dirty padding and blank bytes must not matter
        recs["id"]["pad"] = 0xEE
        recs["metrics"]["pad2"] = 0x55
        recs["metrics"]["pad4"] = 0x77
    want = account(recs, max_entries, dedup)
    got = O.run_accounter(recs, max_entries, 1 if dedup else 0)
    assert [(r, len(b)) for r, b in got] == [(r, len(e)) for r, e in want]
    for (_, batch), (_, entries) in zip(got, want):
        for rec in batch:
            d = entries[rec["id"].tobytes()[:39]]
            g = to_dict(rec["metrics"])
            assert g == d, (rec["id"].tobytes().hex(), {k: (g[k], d[k]) for k in g if g[k] != d[k]})
