// rb_rccl.hpp -- the one exchange step of the sharded renderer: every rank's RGBA8
// stripes to the root device, de-interleaved there into the frame, one read-back.  RCCL (ncclSend / ncclRecv
// in one group = a gather over xGMI, each peer -> root on its own link) is loaded with dlopen the first time a
// multi-device engine is made, so that single-device users need no librccl at all; a peer-copy transport
// (hipMemcpyPeerAsync) exists for hosts without RCCL and for the tests that run several shards on one GPU.
// Not part of the ABI.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <string>
#include <vector>

namespace rb {

struct GatherSource {
    int device;
    hipStream_t stream;          // the part's exchange stream (not the one it renders on: a pass that runs ahead of the
                                 // delivered frame must not hold the gather back)
    hipEvent_t ready;            // recorded behind the launches that produced `rgba`; the exchange stream waits for it
    const uint32_t* rgba;        // its padded stripe buffer (padded_rows * width)
};

struct Gather {
    uint32_t nranks = 0, rank = 0;   // one process per device (rb_comm_init_rank): nranks > 1
    bool group = false;              // several devices in this process (rb_create_multi)
    bool peer_copy = false;          // transport: hipMemcpyPeerAsync instead of RCCL
    std::vector<void*> comms;        // ncclComm_t: one per part (group) or one (process)
    std::vector<int> devices;
    int root_device = 0;
    uint32_t* gathered = nullptr;    // root: [rank][padded_rows][width]
    size_t gathered_words = 0;
    uint32_t* frame = nullptr;       // root: [height][width]
    size_t frame_words = 0;
    std::vector<hipEvent_t> arrived; // peer-copy transport: one per part
    hipEvent_t t0 = nullptr, t1 = nullptr;   // around this rank's share of the last gather (root: receives + de-interleave)
    float last_ms = 0.0f;
};

// all return 0 or a non-zero status with `why` set
int gather_init_group(Gather& g, const std::vector<int>& devices, bool peer_copy, std::string& why);
int gather_available(std::string& why);   // can RCCL be loaded here?  (dlopen + dlsym only: no id, no socket, no thread)
int gather_unique_id(uint8_t* id128, std::string& why);
int gather_init_rank(Gather& g, int device, const uint8_t* id128, uint32_t rank, uint32_t nranks, std::string& why);
int gather_group(Gather& g, const std::vector<GatherSource>& parts, uint32_t width, uint32_t height, uint32_t padded_rows,
                 uint32_t stripe_rows, uint8_t* rgba_out, std::string& why);
int gather_process(Gather& g, const uint32_t* local_rgba, uint32_t width, uint32_t height, uint32_t padded_rows,
                   uint32_t stripe_rows, hipStream_t stream, hipEvent_t ready, uint8_t* rgba_out, std::string& why);
// what RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank of this rank's handle): 0 ranks when
// the exchange does not go through RCCL (one device, or the peer-copy transport)
int gather_comm_info(const Gather& g, uint32_t* rccl_ranks, uint32_t* rccl_rank, std::string& why);
void* gather_frame_ptr(const Gather& g);
void gather_destroy(Gather& g);

}  // namespace rb
