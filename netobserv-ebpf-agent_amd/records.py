"""numpy views of the record ABI (include/nfagg.h; reference bpf/types.h:94-215).

Offsets are spelled out so that a mismatch with the C structs is a test
failure (tests/test_abi.py), not a silent reinterpretation.
"""
import numpy as np

FLOW_ID = np.dtype({
    "names": ["src_ip", "dst_ip", "src_port", "dst_port", "transport_protocol", "icmp_type", "icmp_code", "pad_"],
    "formats": [("u1", 16), ("u1", 16), "<u2", "<u2", "u1", "u1", "u1", "u1"],
    "offsets": [0, 16, 32, 34, 36, 37, 38, 39],
    "itemsize": 40,
})

FLOW_METRICS = np.dtype({
    "names": ["start_mono_time_ts", "end_mono_time_ts", "bytes", "packets", "eth_protocol", "flags",
              "src_mac", "dst_mac", "if_index_first_seen", "lock", "sampling", "direction_first_seen",
              "errno_", "dscp", "nb_observed_intf", "observed_direction", "pad2_", "observed_intf",
              "ssl_version", "tls_cipher_suite", "tls_key_share", "tls_types", "misc_flags", "pad4_"],
    "formats": ["<u8", "<u8", "<u8", "<u4", "<u2", "<u2", ("u1", 6), ("u1", 6), "<u4", "<u4", "<u4", "u1",
                "u1", "u1", "u1", ("u1", 6), ("u1", 2), ("<u4", 6), "<u2", "<u2", "<u2", "u1", "u1", ("u1", 4)],
    "offsets": [0, 8, 16, 24, 28, 30, 32, 38, 44, 48, 52, 56, 57, 58, 59, 60, 66, 68, 92, 94, 96, 98, 99, 100],
    "itemsize": 104,
})

FLOW_RECORD = np.dtype({"names": ["id", "metrics"], "formats": [FLOW_ID, FLOW_METRICS], "offsets": [0, 40], "itemsize": 144})

ADDITIONAL = np.dtype({
    "names": ["start_mono_time_ts", "end_mono_time_ts", "flow_rtt", "ipsec_encrypted_ret", "eth_protocol", "ipsec_encrypted", "pad_"],
    "formats": ["<u8", "<u8", "<u8", "<i4", "<u2", "u1", "u1"], "offsets": [0, 8, 16, 24, 28, 30, 31], "itemsize": 32})

DNS = np.dtype({
    "names": ["start_mono_time_ts", "end_mono_time_ts", "latency", "id", "flags", "eth_protocol", "errno_", "name", "pad_"],
    "formats": ["<u8", "<u8", "<u8", "<u2", "<u2", "<u2", "u1", ("u1", 32), "u1"],
    "offsets": [0, 8, 16, 24, 26, 28, 30, 31, 63], "itemsize": 64})

PKT_DROP = np.dtype({
    "names": ["start_mono_time_ts", "end_mono_time_ts", "bytes", "packets", "latest_drop_cause", "latest_flags", "eth_protocol", "latest_state", "pad_"],
    "formats": ["<u8", "<u8", "<u2", "<u2", "<u4", "<u2", "<u2", "u1", ("u1", 3)],
    "offsets": [0, 8, 16, 18, 20, 24, 26, 28, 29], "itemsize": 32})

NETWORK_EVENTS = np.dtype({
    "names": ["start_mono_time_ts", "end_mono_time_ts", "network_events", "bytes", "packets", "eth_protocol", "network_events_idx", "pad_"],
    "formats": ["<u8", "<u8", ("u1", (4, 8)), ("<u2", 4), ("<u2", 4), "<u2", "u1", ("u1", 5)],
    "offsets": [0, 8, 16, 48, 56, 64, 66, 67], "itemsize": 72})

XLAT = np.dtype({
    "names": ["start_mono_time_ts", "end_mono_time_ts", "saddr", "daddr", "sport", "dport", "zone_id", "eth_protocol"],
    "formats": ["<u8", "<u8", ("u1", 16), ("u1", 16), "<u2", "<u2", "<u2", "<u2"],
    "offsets": [0, 8, 16, 32, 48, 50, 52, 54], "itemsize": 56})

QUIC = np.dtype({
    "names": ["start_mono_time_ts", "end_mono_time_ts", "version", "eth_protocol", "seen_long_hdr", "seen_short_hdr"],
    "formats": ["<u8", "<u8", "<u4", "<u2", "u1", "u1"], "offsets": [0, 8, 16, 20, 22, 23], "itemsize": 24})

ROLLUP_KINDS = {"additional": ADDITIONAL, "dns": DNS, "drops": PKT_DROP, "network_events": NETWORK_EVENTS, "xlat": XLAT, "quic": QUIC}


def sort_by_key(records: np.ndarray) -> np.ndarray:
    """Order evicted records by their 40 key bytes (memcmp order). The reference's
    eviction order is Go map order (random); tests compare by key (account_test.go:94-98)."""
    raw = np.ascontiguousarray(records).view(np.uint8).reshape(-1, 144)
    keys = raw[:, :40]
    order = np.lexsort(tuple(keys[:, c] for c in range(39, -1, -1)))
    return records[order]


# nfagg_intf_name (include/nfagg.h): one row of the interface namer table used by the protobuf encoder
INTF_NAME = np.dtype({"names": ["if_index", "mac", "has_mac", "name_len", "name", "udn_len", "udn"],
                      "formats": ["<u4", ("u1", 6), "u1", "u1", "S16", "u1", "S63"],
                      "offsets": [0, 4, 10, 11, 12, 28, 29], "itemsize": 92})


def intf_table(rows) -> np.ndarray:
    """rows: iterable of (if_index, mac bytes or None, name str, udn str) -> INTF_NAME array."""
    rows = list(rows)
    t = np.zeros(len(rows), dtype=INTF_NAME)
    for k, (ifx, mac, name, udn) in enumerate(rows):
        nb, ub = name.encode(), udn.encode()
        if len(nb) > 16 or len(ub) > 63:
            raise ValueError("interface name <= 16 bytes, udn <= 63 bytes")
        t[k]["if_index"] = ifx
        if mac is not None:
            t[k]["mac"] = np.frombuffer(bytes(mac), dtype=np.uint8)
            t[k]["has_mac"] = 1
        t[k]["name"], t[k]["name_len"] = nb, len(nb)
        t[k]["udn"], t[k]["udn_len"] = ub, len(ub)
    return t
