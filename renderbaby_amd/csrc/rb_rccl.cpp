// rb_rccl.cpp -- see rb_rccl.hpp.
#include "rb_rccl.hpp"

#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "rb_internal.hpp"

namespace rb {
namespace {

// The slice of the RCCL API the gather needs (rccl.h: ncclUniqueId is 128 opaque bytes passed by value,
// ncclUint8 = 1, ncclSuccess = 0).
struct NcclId {
    char bytes[128];
};
struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(NcclId*) = nullptr;
    int (*CommInitRank)(void**, int, NcclId, int) = nullptr;
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*CommUserRank)(void*, int*) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
};
constexpr int kNcclUint8 = 1;

Rccl& rccl() {
    static Rccl api;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy that is already mapped (e.g. the one a PyTorch process brought) wins: two RCCLs in one
        // process would each own a HIP context
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)
            if ((api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!api.handle)
            for (const char* n : names)
                if ((api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!api.handle) {
            const char* de = dlerror();
            api.error = std::string("librccl.so not found: ") + (de ? de : "");
            return;
        }
        auto sym = [&](const char* s) -> void* {
            void* p = dlsym(api.handle, s);
            if (!p && api.error.empty()) api.error = std::string("librccl.so lacks ") + s;
            return p;
        };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
        api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(sym("ncclCommUserRank"));
        api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
        api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return api;
}

bool rccl_ok(std::string& why) {
    Rccl& r = rccl();
    if (!r.error.empty()) {
        why = r.error;
        return false;
    }
    return true;
}

#define NCCL_TRY(call)                                                                      \
    do {                                                                                    \
        const int _st = (call);                                                             \
        if (_st != 0) {                                                                     \
            why = std::string(#call) + " failed: " + rccl().GetErrorString(_st);            \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)
#define HIP_TRY_W(call)                                                                     \
    do {                                                                                    \
        const hipError_t _st = (call);                                                      \
        if (_st != hipSuccess) {                                                            \
            why = std::string(#call) + " failed: " + hipGetErrorString(_st);                \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)

// root-side buffers, (re)sized for the frame at hand
int ensure_root_buffers(Gather& g, uint32_t n, uint32_t width, uint32_t height, uint32_t padded_rows, std::string& why) {
    const size_t gw = static_cast<size_t>(n) * padded_rows * width, fw = static_cast<size_t>(height) * width;
    HIP_TRY_W(hipSetDevice(g.root_device));
    if (gw != g.gathered_words) {
        if (g.gathered) (void)hipFree(g.gathered);
        g.gathered = nullptr;
        g.gathered_words = 0;
        if (gw) HIP_TRY_W(hipMalloc(reinterpret_cast<void**>(&g.gathered), gw * 4));
        g.gathered_words = gw;
    }
    if (fw != g.frame_words) {
        if (g.frame) (void)hipFree(g.frame);
        g.frame = nullptr;
        g.frame_words = 0;
        if (fw) HIP_TRY_W(hipMalloc(reinterpret_cast<void**>(&g.frame), fw * 4));
        g.frame_words = fw;
    }
    return 0;
}

int ensure_timers(Gather& g, std::string& why) {   // on the device that is current
    if (!g.t0) HIP_TRY_W(hipEventCreate(&g.t0));
    if (!g.t1) HIP_TRY_W(hipEventCreate(&g.t1));
    return 0;
}

// own stripes into slot `rank` of the gathered buffer, de-interleave, then the frame to the caller: a DMA on the same
// stream into page-locked memory (rb_host_alloc, or memory the caller registered with HIP), else wait and copy
int assemble_and_read(Gather& g, uint32_t n, uint32_t rank, const uint32_t* own, uint32_t width, uint32_t height,
                      uint32_t padded_rows, uint32_t stripe_rows, hipStream_t stream, uint8_t* rgba_out, std::string& why) {
    const size_t words = static_cast<size_t>(padded_rows) * width;
    if (words && height) {
        HIP_TRY_W(hipMemcpyAsync(g.gathered + rank * words, own, words * 4, hipMemcpyDeviceToDevice, stream));
        const int rc = launch_deinterleave(g.gathered, g.frame, width, height, padded_rows, stripe_rows, n, stream);
        if (rc) {
            why = std::string("de-interleave launch failed: ") + hipGetErrorString(static_cast<hipError_t>(rc));
            return 1;
        }
    }
    if (g.t1) HIP_TRY_W(hipEventRecord(g.t1, stream));
    bool pinned = false;
    if (rgba_out && g.frame_words) {
        hipPointerAttribute_t attr{};
        pinned = hipPointerGetAttributes(&attr, rgba_out) == hipSuccess && attr.type == hipMemoryTypeHost;
        if (!pinned) (void)hipGetLastError();   // an unregistered pointer makes the query fail: the ordinary case
        if (pinned) HIP_TRY_W(hipMemcpyAsync(rgba_out, g.frame, g.frame_words * 4, hipMemcpyDeviceToHost, stream));
    }
    HIP_TRY_W(hipStreamSynchronize(stream));
    if (rgba_out && g.frame_words && !pinned) HIP_TRY_W(hipMemcpy(rgba_out, g.frame, g.frame_words * 4, hipMemcpyDeviceToHost));
    if (g.t0 && g.t1) (void)hipEventElapsedTime(&g.last_ms, g.t0, g.t1);
    return 0;
}

// A failed call between ncclGroupStart and ncclGroupEnd must not leave the thread's group open (every later
// RCCL call of this thread would queue into it): remember the first failure, close the group, then report.
struct GroupScope {
    std::string first;
    bool open = false;
    void fail(const char* what, const std::string& msg) {
        if (first.empty()) first = std::string(what) + " failed: " + msg;
    }
    void nccl(const char* what, int st) {
        if (st != 0) fail(what, rccl().GetErrorString(st));
    }
    void hip(const char* what, hipError_t st) {
        if (st != hipSuccess) fail(what, hipGetErrorString(st));
    }
    bool ok() const { return first.empty(); }
    void start() {
        const int st = rccl().GroupStart();
        if (st != 0) fail("ncclGroupStart", rccl().GetErrorString(st));
        else open = true;
    }
    int end(std::string& why) {   // 0, or 1 with `why`
        if (open) {
            open = false;
            const int st = rccl().GroupEnd();
            if (st != 0) fail("ncclGroupEnd", rccl().GetErrorString(st));
        }
        if (first.empty()) return 0;
        why = first;
        return 1;
    }
};

}  // namespace

int gather_init_group(Gather& g, const std::vector<int>& devices, bool peer_copy, std::string& why) {
    g.group = true;
    g.devices = devices;
    g.root_device = devices[0];
    g.peer_copy = peer_copy;
    const int n = static_cast<int>(devices.size());
    if (n == 1) return 0;   // nothing to exchange
    if (peer_copy) {
        g.arrived.resize(n, nullptr);
        for (int r = 1; r < n; ++r) {
            HIP_TRY_W(hipSetDevice(devices[r]));
            HIP_TRY_W(hipEventCreateWithFlags(&g.arrived[r], hipEventDisableTiming));
        }
        return 0;
    }
    if (!rccl_ok(why)) return 1;
    g.comms.assign(n, nullptr);
    NCCL_TRY(rccl().CommInitAll(g.comms.data(), n, devices.data()));   // single process, n devices
    return 0;
}

int gather_available(std::string& why) { return rccl_ok(why) ? 0 : 1; }

int gather_unique_id(uint8_t* id128, std::string& why) {
    if (!rccl_ok(why)) return 1;
    NcclId id;
    NCCL_TRY(rccl().GetUniqueId(&id));
    std::memcpy(id128, id.bytes, 128);
    return 0;
}

int gather_init_rank(Gather& g, int device, const uint8_t* id128, uint32_t rank, uint32_t nranks, std::string& why) {
    if (nranks == 0 || rank >= nranks) {
        why = "bad communicator rank";
        return 1;
    }
    if (!rccl_ok(why)) return 1;
    NcclId id;
    std::memcpy(id.bytes, id128, 128);
    g.comms.assign(1, nullptr);
    HIP_TRY_W(hipSetDevice(device));
    NCCL_TRY(rccl().CommInitRank(&g.comms[0], static_cast<int>(nranks), id, static_cast<int>(rank)));
    g.nranks = nranks;
    g.rank = rank;
    g.root_device = device;
    g.devices.assign(1, device);
    return 0;
}

int gather_group(Gather& g, const std::vector<GatherSource>& parts, uint32_t width, uint32_t height, uint32_t padded_rows,
                 uint32_t stripe_rows, uint8_t* rgba_out, std::string& why) {
    const uint32_t n = static_cast<uint32_t>(parts.size());
    if (ensure_root_buffers(g, n, width, height, padded_rows, why)) return 1;
    if (ensure_timers(g, why)) return 1;
    const size_t words = static_cast<size_t>(padded_rows) * width;
    // every part's exchange stream starts behind the launches that produced its stripes
    for (uint32_t r = 0; r < n; ++r) {
        HIP_TRY_W(hipSetDevice(parts[r].device));
        if (parts[r].ready) HIP_TRY_W(hipStreamWaitEvent(parts[r].stream, parts[r].ready, 0));
    }
    HIP_TRY_W(hipSetDevice(g.root_device));
    HIP_TRY_W(hipEventRecord(g.t0, parts[0].stream));
    if (n > 1 && words) {
        if (g.peer_copy) {
            for (uint32_t r = 1; r < n; ++r) {
                HIP_TRY_W(hipSetDevice(parts[r].device));
                if (parts[r].device == g.root_device)
                    HIP_TRY_W(hipMemcpyAsync(g.gathered + r * words, parts[r].rgba, words * 4, hipMemcpyDeviceToDevice, parts[r].stream));
                else
                    HIP_TRY_W(hipMemcpyPeerAsync(g.gathered + r * words, g.root_device, parts[r].rgba, parts[r].device, words * 4,
                                                 parts[r].stream));
                HIP_TRY_W(hipEventRecord(g.arrived[r], parts[r].stream));
            }
            HIP_TRY_W(hipSetDevice(g.root_device));
            for (uint32_t r = 1; r < n; ++r) HIP_TRY_W(hipStreamWaitEvent(parts[0].stream, g.arrived[r], 0));
        } else {
            // one grouped call = the gather: every peer sends its stripes on its exchange stream, the root posts the
            // matching receives on its own
            GroupScope grp;
            grp.start();
            for (uint32_t r = 1; r < n && grp.ok(); ++r) {
                grp.hip("hipSetDevice", hipSetDevice(parts[r].device));
                if (grp.ok()) grp.nccl("ncclSend", rccl().Send(parts[r].rgba, words * 4, kNcclUint8, 0, g.comms[r], parts[r].stream));
            }
            if (grp.ok()) grp.hip("hipSetDevice", hipSetDevice(g.root_device));
            for (uint32_t r = 1; r < n && grp.ok(); ++r)
                grp.nccl("ncclRecv", rccl().Recv(g.gathered + r * words, words * 4, kNcclUint8, static_cast<int>(r), g.comms[0], parts[0].stream));
            if (grp.end(why)) return 1;
        }
    }
    HIP_TRY_W(hipSetDevice(g.root_device));
    return assemble_and_read(g, n, 0, parts[0].rgba, width, height, padded_rows, stripe_rows, parts[0].stream, rgba_out, why);
}

int gather_process(Gather& g, const uint32_t* local_rgba, uint32_t width, uint32_t height, uint32_t padded_rows,
                   uint32_t stripe_rows, hipStream_t stream, hipEvent_t ready, uint8_t* rgba_out, std::string& why) {
    const size_t words = static_cast<size_t>(padded_rows) * width;
    if (ensure_timers(g, why)) return 1;
    if (ready) HIP_TRY_W(hipStreamWaitEvent(stream, ready, 0));
    HIP_TRY_W(hipEventRecord(g.t0, stream));
    if (g.rank != 0) {
        if (words) NCCL_TRY(rccl().Send(local_rgba, words * 4, kNcclUint8, 0, g.comms[0], stream));
        HIP_TRY_W(hipEventRecord(g.t1, stream));
        HIP_TRY_W(hipStreamSynchronize(stream));
        (void)hipEventElapsedTime(&g.last_ms, g.t0, g.t1);
        return 0;
    }
    if (ensure_root_buffers(g, g.nranks, width, height, padded_rows, why)) return 1;
    if (words && g.nranks > 1) {
        GroupScope grp;
        grp.start();
        for (uint32_t r = 1; r < g.nranks && grp.ok(); ++r)
            grp.nccl("ncclRecv", rccl().Recv(g.gathered + r * words, words * 4, kNcclUint8, static_cast<int>(r), g.comms[0], stream));
        if (grp.end(why)) return 1;
    }
    return assemble_and_read(g, g.nranks, 0, local_rgba, width, height, padded_rows, stripe_rows, stream, rgba_out, why);
}

int gather_comm_info(const Gather& g, uint32_t* rccl_ranks, uint32_t* rccl_rank, std::string& why) {
    if (rccl_ranks) *rccl_ranks = 0;
    if (rccl_rank) *rccl_rank = 0;
    if (g.comms.empty() || !g.comms[0]) return 0;
    int n = 0, r = 0;
    NCCL_TRY(rccl().CommCount(g.comms[0], &n));
    NCCL_TRY(rccl().CommUserRank(g.comms[0], &r));
    if (rccl_ranks) *rccl_ranks = static_cast<uint32_t>(n);
    if (rccl_rank) *rccl_rank = static_cast<uint32_t>(r);
    return 0;
}

void* gather_frame_ptr(const Gather& g) { return g.frame; }

void gather_destroy(Gather& g) {
    for (size_t i = 0; i < g.comms.size(); ++i)
        if (g.comms[i]) {
            if (i < g.devices.size()) (void)hipSetDevice(g.devices[i]);
            (void)rccl().CommDestroy(g.comms[i]);
        }
    g.comms.clear();
    for (size_t r = 0; r < g.arrived.size(); ++r)
        if (g.arrived[r]) {
            (void)hipSetDevice(g.devices[r]);
            (void)hipEventDestroy(g.arrived[r]);
        }
    g.arrived.clear();
    if (g.gathered || g.frame || g.t0 || g.t1) (void)hipSetDevice(g.root_device);
    if (g.t0) (void)hipEventDestroy(g.t0);
    if (g.t1) (void)hipEventDestroy(g.t1);
    g.t0 = g.t1 = nullptr;
    if (g.gathered) (void)hipFree(g.gathered);
    if (g.frame) (void)hipFree(g.frame);
    g.gathered = g.frame = nullptr;
    g.gathered_words = g.frame_words = 0;
    g.nranks = 0;
}

}  // namespace rb
