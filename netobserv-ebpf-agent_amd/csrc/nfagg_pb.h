// nfagg_pb.h — launch interface of the record -> protobuf kernels (nfagg_pb.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nfagg.h"

namespace nfagg {

struct PbParams {
    int64_t now_sec, now_nsec;    // currentTime, normalised (0 <= nsec < 1e9)
    uint64_t mono_now;
    uint32_t agent_ip_w[4];       // Record.AgentIP as a 16-byte net.IP, four little-endian dwords
    uint32_t agent_is_v4;         // net.IP.To4() != nil
    const nfagg_intf_name* names; // device copy of the namer table, stably sorted by if_index
    uint32_t n_names;
    uint32_t unknown_len;
    char unknown[16];
};

// Per-flow feature parts of model.BpfFlowContent (the MapTracer branch), DEVICE pointers, struct-of-arrays
// indexed like the records; a part is present for record i when its array is non-null and present[i]
// carries its bit (1 << rollup kind).
struct PbFeat {
    const uint8_t* present = nullptr;
    const uint8_t* additional = nullptr;   // nfagg_additional_metrics[n]   32 B
    const uint8_t* dns = nullptr;          // nfagg_dns_metrics[n]          64 B
    const uint8_t* drops = nullptr;        // nfagg_pkt_drop_metrics[n]     32 B
    const uint8_t* xlat = nullptr;         // nfagg_xlat_metrics[n]         56 B
    const uint8_t* quic = nullptr;         // nfagg_quic_metrics[n]         24 B
};

hipError_t launch_pb_size(const void* d_recs, uint64_t n, const PbParams& P, const PbFeat& F, uint32_t* d_body_len, uint32_t* d_local_off,
                          uint32_t* d_block_sum, uint64_t* d_block_base, hipStream_t s);
hipError_t launch_pb_write(const void* d_recs, uint64_t n, const PbParams& P, const PbFeat& F, const uint32_t* d_body_len, const uint32_t* d_local_off,
                           const uint64_t* d_block_base, void* d_out, uint64_t* d_frame_offsets, void* d_kafka_keys, uint64_t total_bytes, hipStream_t s);

}  // namespace nfagg
