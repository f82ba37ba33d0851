// rb_kernels.hip -- gfx950 (CDNA4, wave64) kernels for RenderBaby's path-tracing
// hot path: the per-pixel / per-sample loop of
// crates/engine-pathtracer/src/shader.wgsl (main :673-723, trace_ray :522-662,
// intersect_bvh :282-392, intersect_* :193-280,402-414,664-671, PCG :417-446,
// scatter :459-490, sample_texture :153-191, color_map :137-151).
//
// Numerics contract (bit-exact against oracle/rb_oracle.c; DESIGN.md "Numerics"):
//   every f32 operation is a single IEEE binary32 op, round-to-nearest-even, no FMA
//   contraction (this TU is built with -ffp-contract=off and the pragma below),
//   correctly-rounded / and sqrt (-fhip-fp32-correctly-rounded-divide-sqrt),
//   subnormals kept (hipcc default float mode), dot = (x*x' + y*y') + z*z'.
//
// No MFMA: the work is branchy traversal and 3-wide dot products.  What matters
// here is lane utilisation (path regeneration keeps all 64 lanes on the
// intersection code), scalar/LDS residency of the scene, and 16-byte loads.
#include "rb_device_shade.hpp"

#pragma clang fp contract(off)

namespace rb {
namespace {

// This lane's column of the block's traversal stacks (rb_internal.hpp, kStackEntryBytes): entry k at column[k * block].
template <class Entry>
DEV Entry* stack_column(Entry* s_stack, uint32_t tid) {
    static_assert(sizeof(Entry) == kStackEntryBytes, "the traversal stacks are columns of 4-byte entries shared by every walk of a kernel");
    return &s_stack[tid];
}

// ======================================================= kernel: PIXEL ====
// One thread per pixel, 8x8 pixels per wavefront, nested sample / bounce loops:
// the shape of the reference's dispatch (one invocation = one pixel,
// gpu_wrapper.rs:380-384) with all passes of a launch folded into the kernel.
constexpr uint32_t kPixelBlock = 256;

template <bool STATS>
__global__ void __launch_bounds__(kPixelBlock) k_pixel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_stack[];  // [stack_depth][blockDim.x]; empty for single-node trees
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    const uint32_t tiles_x = (p.u.width + 15u) / 16u;
    const uint32_t bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    const uint32_t x = bx * 16u + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t ly = by * 16u + (wave >> 1) * 8u + (lane >> 3);
    Tally<STATS> tl;
    bool active = (x < p.u.width) && (ly < p.local_rows);
    uint32_t y = 0;
    if (active) {
        y = global_row(p, ly);
        active = y < p.u.height;
    }
    if (active) {
        const uint32_t pixel_index = y * p.u.width + x;
        const float4 a4 = reinterpret_cast<const float4*>(p.accum_in)[(size_t)ly * p.u.width + x];
        f3 acc = mk(a4.x, a4.y, a4.z);
        uint32_t total = f2u(a4.w);
        for (uint32_t pass = p.first_pass; pass < p.first_pass + p.n_passes; pass++) {
            for (uint32_t s = 0; s < p.samples_per_pass; s++) {
                Path pt;
                start_path(fresh_params(p), x, y, pixel_index, pass * p.samples_per_pass + s, pt);
                if (p.u.max_depth > 0u) {
                    while (segment<STATS>(p, pt, &s_stack[tid], kPixelBlock, tl)) {
                    }
                }
                tl.paths++;
                acc = acc + pt.color;
                total = total + 1u;
            }
        }
        store_pixel(p, x, ly, acc, total);
    }
    flush_tally<STATS>(tl, p.counters);
}

// ======================================================= kernel: QUEUE ====
// Persistent wavefronts.  Each lane owns one pixel at a time and walks its
// samples in order (so the f32 accumulation order is the reference's); a lane
// whose path ends starts its next sample immediately (path regeneration), and a
// lane whose pixel is finished takes the next pixel from a global queue: the
// wave ballots the idle lanes, lane 0 reserves popcount(idle) pixels with one
// atomic, and each idle lane picks its slot by prefix count.  All live lanes
// therefore execute the intersection code together on every iteration.
constexpr uint32_t kQueueBlock = 256;

template <bool STATS>
__global__ void __launch_bounds__(kQueueBlock) k_queue(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_stack[];  // [stack_depth][blockDim.x]; empty for single-node trees
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t width = p.u.width;
    const uint32_t tiles_x = (width + 7u) / 8u;
    const uint32_t tiles_y = (p.local_rows + 7u) / 8u;
    const uint32_t total_items = tiles_x * tiles_y * 64u;  // host guarantees < 2^32
    const uint32_t samples_total = p.n_passes * p.samples_per_pass;
    const uint32_t sample_base = p.first_pass * p.samples_per_pass;
    Tally<STATS> tl;

    bool have_pixel = false;   // lane owns a pixel
    bool exhausted = false;    // queue is empty for this wave
    uint32_t x = 0, ly = 0, y = 0, pixel_index = 0, sample = 0, total = 0;
    f3 acc = mk(0, 0, 0);
    Path pt;
    pt.depth = 0;
    bool in_path = false;

    for (;;) {
        // ---- refill idle lanes from the queue
        if (!exhausted) {
            const unsigned long long idle = __ballot(!have_pixel);
            if (idle != 0ull) {
                const uint32_t n = (uint32_t)__popcll(idle);
                uint32_t base = 0;
                if (lane == (uint32_t)(__ffsll((long long)idle) - 1)) base = atomicAdd(p.queue, n);
                base = __shfl(base, __ffsll((long long)idle) - 1, 64);
                if (!have_pixel) {
                    const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                    const uint32_t item = base + rank;
                    if (item < total_items) {
                        const uint32_t tile = item >> 6, in = item & 63u;
                        x = (tile % tiles_x) * 8u + (in & 7u);
                        ly = (tile / tiles_x) * 8u + (in >> 3);
                        bool ok = (x < width) && (ly < p.local_rows);
                        if (ok) {
                            y = global_row(p, ly);
                            ok = y < p.u.height;
                        }
                        if (ok) {
                            pixel_index = y * width + x;
                            const float4 a4 = reinterpret_cast<const float4*>(p.accum_in)[(size_t)ly * width + x];
                            acc = mk(a4.x, a4.y, a4.z);
                            total = f2u(a4.w);
                            sample = 0;
                            have_pixel = true;
                            in_path = false;
                        }
                    }
                }
                if (base + n >= total_items) exhausted = true;
            }
        }
        if (__ballot(have_pixel) == 0ull) {
            if (exhausted) break;
            continue;
        }
        if (have_pixel) {
            if (!in_path) {
                start_path(fresh_params(p), x, y, pixel_index, sample_base + sample, pt);
                in_path = p.u.max_depth > 0u;
            }
            if (in_path) in_path = segment<STATS>(p, pt, &s_stack[tid], kQueueBlock, tl);
            if (!in_path) {
                tl.paths++;
                acc = acc + pt.color;
                total = total + 1u;
                sample++;
                if (sample == samples_total) {
                    store_pixel(p, x, ly, acc, total);
                    have_pixel = false;
                }
            }
        }
    }
    flush_tally<STATS>(tl, p.counters);
}

// ============================================== kernels: TRACE + ACCUMULATE ==
// Two-phase form of the same computation.  Phase 1 (k_trace): persistent
// wavefronts pull single (pixel, sample) items from a global queue -- a ballot of
// the idle lanes, one atomic by the first idle lane for popcount(idle) items, a
// prefix count to hand them out -- trace the path and store its radiance (16 B)
// into an HBM buffer laid out [tile][sample][64 pixels].  A lane that finishes a
// path regenerates at once, so all 64 lanes stay on the intersection code and
// the launch tail is one path, not one pixel's worth of samples.  Nothing in
// phase 1 depends on order.  Phase 2 (k_accumulate): one lane per pixel adds the
// samples to the accumulation IN SAMPLE ORDER (f32 addition is not associative:
// this is what keeps the frame bit-identical to the reference order,
// shader.wgsl:712-717), then tone-maps and packs (:720-722).  Phase 2 is a
// coalesced 1 KiB-per-wave stream and is HBM-bound; phase 1 is ALU-bound.
constexpr uint32_t kTraceBlock = 256;

// One path's radiance -> its (pixel, sample) slot of the colour buffer, read once by k_accumulate.
// Lanes finish at different times, so these are lone 16-byte stores: HBM takes them as 32-byte
// writes (measured 2.2x the payload).  Streaming stores at least avoid the line fills and the early
// partial write-backs of cached ones (2.45x plus 0.2x fetched); the kernel time is the same either way.
// k_trace, whose frames are the large ones, does not store this way: see ColorRing.
#ifndef RB_COLOR_STORE_NT
#define RB_COLOR_STORE_NT 1
#endif
DEV void store_color4(float4* __restrict__ colors, uint32_t item, v4f v) {
#if RB_COLOR_STORE_NT
    __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(colors) + item);
#else
    reinterpret_cast<v4f*>(colors)[item] = v;
#endif
}
DEV void store_color(float4* __restrict__ colors, uint32_t item, f3 c) {
    const v4f v = {c.x, c.y, c.z, 0.0f};
    store_color4(colors, item, v);
}

// Write combining for the colour buffer of k_trace: one ring of kRingRows x 64 entries per wavefront in LDS.
// A wave is handed items in ascending order from reservations that are multiples of 64, so it works through
// 64-item rows of the buffer (one sample of one 8 x 8 tile: 1 KiB); row r uses ring slice r mod kRingRows, entry
// item mod 64.  A finished path parks its radiance there, tagged item + 1 (one ds_write_b128).  Before the first
// item of a new row is handed out, the whole wave empties the slice that row will use: lane l stores entry l, so
// the two (usually all eight) finished neighbours of a 32-byte sector (128-byte line) leave in one instruction
// and the memory side sees whole sectors instead of lone 16-byte stores that it rounds up to 32 bytes each.
// Reservations are multiples of kRingRows rows (launch_render), so the slices come round in order whatever the
// jumps between reservations: that slice holds the row handed out kRingRows rows -- about 20 loop iterations -- earlier, and a path takes at
// most max_depth iterations, so at the default depths nearly every entry is there.  A lane still tracing an item of that
// slice is marked (kDirect in its item word) and stores on its own when it finishes, as every lane used to.
// Nothing is ever parked in an occupied entry: the slice was emptied before its row's first item went out, and
// the stragglers of the previous row were marked in the same step.
constexpr uint32_t kRingRows = 4u, kDirect = 0x80000000u;
// LDS-qualified pointers: with generic ones the compiler folds "park or store directly" into one FLAT store through a
// selected address, which is slower than either and loses the streaming hint
typedef __attribute__((address_space(3))) v4f lds_v4f;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
struct ColorRing {
    lds_v4f* ring;  // this wave's kRingRows * 64 entries

    DEV ColorRing(v4f* block_rings, uint32_t wave, uint32_t lane)
        : ring((lds_v4f*)(block_rings + wave * (kRingRows * 64u))) {
        for (uint32_t k = 0; k < kRingRows; k++) tag(k * 64u + lane) = 0u;
    }
    DEV lds_u32& tag(uint32_t e) const { return ((lds_u32*)ring)[e * 4u + 3u]; }

    // a finished path; `item` may carry kDirect
    DEV void finish(float4* __restrict__ colors, uint32_t item, f3 c) const {
        if (item & kDirect) {
            store_color(colors, item & ~kDirect, c);
        } else {
            const v4f v = {c.x, c.y, c.z, __uint_as_float(item + 1u)};
            ring[item & (kRingRows * 64u - 1u)] = v;
        }
    }
    // all 64 lanes, before the first item of row `row` is handed out
    DEV void open_row(float4* __restrict__ colors, uint32_t lane, uint32_t row, bool active, uint32_t& item) const {
        const uint32_t slice = row & (kRingRows - 1u);
        drain(colors, lane, slice);
        if (active && ((item >> 6) & (kRingRows - 1u)) == slice) item |= kDirect;
    }
    DEV void drain(float4* __restrict__ colors, uint32_t lane, uint32_t slice) const {
        const uint32_t e = slice * 64u + lane;
        v4f v = ring[e];
        const uint32_t t = __float_as_uint(v.w);
        if (t != 0u) {
            v.w = 0.0f;
            store_color4(colors, t - 1u, v);
            tag(e) = 0u;
        }
    }
};
#ifndef RB_TRACE_WAVES
#define RB_TRACE_WAVES 6
#endif
#ifndef RB_TRACE_WAVES_BIG
#define RB_TRACE_WAVES_BIG 8
#endif
#ifndef RB_XCD_BANDS
#define RB_XCD_BANDS 1
#endif
struct QueueInit {
    uint32_t start[kQueueGroups];
};
__global__ void k_queue_init(uint32_t* queue, QueueInit qi) {
    if (threadIdx.x < kQueueGroups) queue[threadIdx.x * kQueueStride] = qi.start[threadIdx.x];
}
constexpr uint64_t kTraceManyItems = 3ull << 23;  // launches from here on take k_trace's 8-wave instantiation

// The (pixel, sample) work queue of the stream kernels, one instance per wavefront.  The global
// queue word is touched once per `batch` items (one word sustains only ~90 M atomics/s chip-wide);
// inside a batch the wave hands items to its idle lanes with ballot + prefix count, no memory
// traffic.  `is_idle()` is the calling kernel's notion of an idle lane, `on_item(item, x, y,
// sample_hash)` starts a path on the calling lane (padding pixels of edge tiles are skipped here).
struct ItemQueue {
    uint32_t loc_next = 0, loc_end = 0;  // this wave's reserved item range (wave-uniform)
    uint32_t batch;                      // items per reservation
    uint32_t grp = 0, tried = 0;         // the band this wave draws from, and how many bands it has found empty
    bool exhausted = false;
    // The first reservation is the wave's own: wave w starts on items [w * batch, (w + 1) * batch) and the queue word
    // starts at waves * batch (launch_render), so a launch does not begin with every wave queueing for the one word
    // (4096-8192 atomics at ~90 M/s: up to 45-90 us before the last wave had work).
    // With p.queue_groups = 8 the items are cut into stripes of p.queue_region items (a few tile rows) and dealt to 8 bands, stripe
    // k to band k % 8, one queue word per band: the blocks that share an XCD (blockIdx % 8, MI355X_MICROARCH.md: blocks are dealt
    // round-robin over the XCDs; speed only) work through one band, so that an XCD's L2 holds the part of the tree its stripe looks
    // at, and move on to the other bands when theirs is done.  Interleaved rather than 8 contiguous bands: a frame whose top costs
    // less than its bottom would leave the chip unevenly loaded until the stealing starts.  A band's queue word counts in the band's
    // own index space (stripe after stripe); band_item turns that into the item.
    DEV ItemQueue(const KParams& p, uint32_t total_items) : batch(p.queue_batch) {
        const uint32_t wpb = blockDim.x >> 6;
        if (p.queue_groups <= 1u) {
            const uint64_t first = (uint64_t)(blockIdx.x * wpb + (threadIdx.x >> 6)) * batch;
            if (first < total_items) {
                loc_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)first);
                loc_end = (total_items - loc_next < batch) ? total_items : loc_next + batch;
            }
        } else {
            grp = blockIdx.x % p.queue_groups;
            const uint64_t first = (uint64_t)((blockIdx.x / p.queue_groups) * wpb + (threadIdx.x >> 6)) * batch;
            uint32_t it = 0;
            if (first < 0xFFFFFFFFull && band_item(p, grp, (uint32_t)first, total_items, it)) {
                loc_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)it);
                loc_end = (total_items - loc_next < batch) ? total_items : loc_next + batch;
            }
        }
    }

    DEV bool drained() const { return exhausted && loc_next == loc_end; }

    // position `local` of band g -> item; false beyond the band's last item.  Stripes are whole multiples of the batch, so a
    // reservation never straddles two.
    DEV static bool band_item(const KParams& p, uint32_t g, uint32_t local, uint32_t total_items, uint32_t& item) {
        const uint32_t stripe = p.queue_region;
        const uint32_t k = local / stripe;
        const uint64_t it = ((uint64_t)k * p.queue_groups + g) * stripe + (local - k * stripe);
        if (it >= total_items) return false;
        item = (uint32_t)it;
        return true;
    }

    // a new reservation; false when there is nothing left anywhere
    DEV bool reserve(const KParams& p, uint32_t lane, uint32_t total_items) {
        if (p.queue_groups <= 1u) {
            uint32_t b = 0;
            if (lane == 0u) b = atomicAdd(p.queue, batch);
            b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
            if (b >= total_items) return false;
            loc_next = b;
            loc_end = (total_items - b < batch) ? total_items : b + batch;
            return true;
        }
        while (tried < p.queue_groups) {
            uint32_t b = 0;
            if (lane == 0u) b = atomicAdd(p.queue + grp * kQueueStride, batch);
            b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
            uint32_t it = 0;
            if (band_item(p, grp, b, total_items, it)) {
                loc_next = it;
                loc_end = (total_items - it < batch) ? total_items : it + batch;
                return true;
            }
            grp = (grp + 1u == p.queue_groups) ? 0u : grp + 1u;   // this band is done for good: the words only grow
            tried++;
        }
        return false;
    }

    template <class IsIdle, class OnItem>
    DEV void refill(const KParams& p, uint32_t lane, uint32_t total_items, uint32_t S, uint32_t tiles_x,
                    uint32_t sample_base, IsIdle&& is_idle, OnItem&& on_item) {
        unsigned long long idle = __ballot(is_idle());
        for (int round = 0; round < 2 && idle != 0ull; round++) {
            if (loc_next == loc_end) {
                if (exhausted) break;
                if (!reserve(p, lane, total_items)) {
                    exhausted = true;
                    break;
                }
            }
            const ItemRows rows = item_rows(p, loc_next, S, tiles_x, sample_base);
            const uint32_t avail = loc_end - loc_next;
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            const bool take = ((idle >> lane) & 1ull) != 0ull && rank < avail;
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            const uint32_t taken = n_idle < avail ? n_idle : avail;
            if (take) {
                uint32_t x = 0, y = 0, sample_hash = 0;
                if (item_pixel(p, rows, rank, x, y, sample_hash)) on_item(loc_next + rank, x, y, sample_hash);
            }
            loc_next += taken;
            idle = __ballot(is_idle());
            if (taken == n_idle) break;  // everyone who asked was served (or got a padding item)
        }
    }
};

// MULTI = false: trees of at most one node and at most 64 spheres only (the default instantiation: launch_render sends every
// multi-node tree and every sphere tree to a stepped kernel); MULTI = true: the per-segment ablation of those (opt no_leaf_stepping).
// WAVES = resident waves per SIMD the register allocation aims for: 6 (80 registers, nothing spilled) except for the
// large launches of the default instantiation, which run 3 % faster with 8 (64 registers, 9 spilled outside the
// segment code) -- and 2-3 % slower on one-sample frames, hence two instantiations.
template <bool STATS, bool MULTI, int WAVES = RB_TRACE_WAVES>
__global__ void __launch_bounds__(kTraceBlock, WAVES) k_trace(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_stack[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t tiles_x = (p.u.width + 7u) / 8u;
    const uint32_t tiles_y = (p.local_rows + 7u) / 8u;
    const uint32_t S = p.n_passes * p.samples_per_pass;
    const uint32_t total_items = tiles_x * tiles_y * S * 64u;  // host keeps this < 2^31
    float4* __restrict__ colors = reinterpret_cast<float4*>(p.colors);
    Tally<STATS> tl;
    __shared__ __attribute__((aligned(16))) v4f s_ring[(kTraceBlock / 64u) * kRingRows * 64u];
    ColorRing cring(s_ring, tid >> 6, lane);
    // rows follow each other through the ring's slices only if every reservation starts on a multiple of
    // kRingRows rows; the launcher arranges that for frames large enough, the others store directly
    const uint32_t direct_mask = (p.queue_batch % (kRingRows * 64u)) ? kDirect : 0u;

    bool active = false, exhausted = false;
    uint32_t item = 0;
    uint32_t loc_next = 0, loc_end = 0;  // this wave's reserved item range (wave-uniform)
    const uint32_t batch = p.queue_batch;
    {   // the first reservation is the wave's own (see ItemQueue)
        const uint64_t first = (uint64_t)(blockIdx.x * (kTraceBlock / 64u) + (tid >> 6)) * batch;
        if (first < total_items) {
            loc_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)first);
            loc_end = (total_items - loc_next < batch) ? total_items : loc_next + batch;
        }
    }
    Path pt;
    pt.depth = 0;

    for (;;) {
        // ---- hand items to idle lanes: ItemQueue::refill spelled out (in this kernel, the hottest one,
        // the helper form changes the register allocation and costs 0.8 %)
        unsigned long long idle = __ballot(!active);
        for (int round = 0; round < 2 && idle != 0ull; round++) {
            if (loc_next == loc_end) {
                if (exhausted) break;
                uint32_t b = 0;
                if (lane == 0u) b = atomicAdd(p.queue, batch);
                b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
                if (b >= total_items) {
                    exhausted = true;
                    break;
                }
                loc_next = b;
                loc_end = (total_items - b < batch) ? total_items : b + batch;
            }
            // the launch constants this round needs, fetched again from the kernel arguments (scalar loads): held in
            // scalar registers across the segment code they would be spilled and come back through v_readlane, one
            // vector-unit slot each
            const KParams& rp = fresh_params(p);
            const uint32_t rS = rp.n_passes * rp.samples_per_pass, r_tiles_x = (rp.u.width + 7u) / 8u;
            const uint32_t r_sample_base = rp.first_pass * rp.samples_per_pass, r_width = rp.u.width;
            const ItemRows rows = item_rows(rp, loc_next, rS, r_tiles_x, r_sample_base);
            const uint32_t avail = loc_end - loc_next;
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            const bool take = !active && rank < avail;
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            const uint32_t taken = n_idle < avail ? n_idle : avail;
            if (direct_mask == 0u) {
                if (rows.in0 == 0u) cring.open_row(colors, lane, loc_next >> 6, active, item);
                else if (rows.in0 + taken > 64u) cring.open_row(colors, lane, (loc_next >> 6) + 1u, active, item);
            }
            if (take) {
                const uint32_t it = loc_next + rank;
                uint32_t x = 0, y = 0, sample_hash = 0;
                if (item_pixel(rp, rows, rank, x, y, sample_hash)) {
                    start_path_hashed(rp, x, y, y * r_width + x, sample_hash, pt);
                    item = it | direct_mask;
                    if (p.u.max_depth > 0u) {
                        active = true;
                    } else {
                        colors[it] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        tl.paths++;
                    }
                }
            }
            loc_next += taken;
            idle = __ballot(!active) & ~0ull;
            if (taken == n_idle) break;  // everyone who asked was served (or got a padding item)
        }
        if (__ballot(active) == 0ull) {
            if (exhausted && loc_next == loc_end) break;
            continue;
        }
        if (active) {
            const bool alive = segment<STATS, MULTI>(p, pt, &s_stack[tid], kTraceBlock, tl);
            if (!alive) {
                cring.finish(colors, item, pt.color);
                tl.paths++;
                active = false;
            }
        }
    }
    for (uint32_t k = 0; k < kRingRows; k++) cring.drain(colors, lane, k);
    flush_tally<STATS>(tl, p.counters);
}

// k_trace for multi-node BVHs.  With 128-triangle leaves and no t-culling (the reference's
// traversal, kept for exactness) a ray visits a handful of leaves, but the number varies a lot
// between rays; run per segment, the wavefront waits for its slowest lane (measured ~1/3 lane
// utilisation on the 50k-triangle scene).  Here the scheduling unit is ONE LEAF: every loop
// iteration each traversing lane advances its own stack to its next leaf (node phase) and tests
// that leaf's triangles (leaf phase); a lane whose stack is empty finishes its segment (spheres,
// lights, shading) and starts the next segment or a new path at once, while the other lanes keep
// walking.  Visit order per ray is unchanged, so the winner is the same triangle.
#ifndef RB_BVH_WAVES
#define RB_BVH_WAVES 1
#endif
// LDS = true: the whole tree (48 B per node) and the first 48 B of every prepared triangle are
// first staged into LDS with coalesced 16-byte loads by the whole block, and the walk then reads
// them with ds_read_b128 instead of going through L1/L2 -- for meshes small enough to fit next to
// the traversal stacks: BLOCK = 1024, one block per CU, 16 waves share one staged copy of up to
// 160 KiB (about 2 800 triangles).  1.2-1.6x the L1/L2 path on 160..2 700-triangle meshes.
template <bool STATS, bool LDS, uint32_t BLOCK>
__global__ void __launch_bounds__(BLOCK, RB_BVH_WAVES) k_trace_bvh(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_stack[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t width = p.u.width;
    const uint32_t tiles_x = (width + 7u) / 8u;
    const uint32_t tiles_y = (p.local_rows + 7u) / 8u;
    const uint32_t S = p.n_passes * p.samples_per_pass;
    const uint32_t total_items = tiles_x * tiles_y * S * 64u;
    const uint32_t sample_base = p.first_pass * p.samples_per_pass;
    const uint32_t node_count = p.u.bvh_node_count;
    float4* __restrict__ colors = reinterpret_cast<float4*>(p.colors);
    const cf4p nodes = (cf4p)p.nodes;
    const cf4p ptris = (cf4p)p.ptris;
    uint32_t* const stack = stack_column(s_stack, tid);
    Tally<STATS> tl;

    v4f* const lds_nodes = reinterpret_cast<v4f*>(s_stack + p.stack_depth * BLOCK);
    v4f* const lds_tris = lds_nodes + node_count * 3u;
    if constexpr (LDS) {
        const v4f* gn = reinterpret_cast<const v4f*>(p.nodes);
        const v4f* gt = reinterpret_cast<const v4f*>(p.ptris);
        for (uint32_t i = tid; i < node_count * 3u; i += BLOCK) lds_nodes[i] = gn[i];
        for (uint32_t i = tid; i < p.index_len * 3u; i += BLOCK) {
            const uint32_t slot = i / 3u, part = i - slot * 3u;
            lds_tris[i] = gt[slot * 4u + part];
        }
        __syncthreads();
    }
    auto node_q = [&](uint32_t idx, uint32_t part) -> v4f {
        if constexpr (LDS) return lds_nodes[idx * 3u + part];
        else return nodes[idx * 3u + part];
    };
    auto tri_q = [&](uint32_t slot, uint32_t part) -> v4f {
        if constexpr (LDS) return lds_tris[slot * 3u + part];
        else return ptris[slot * 4u + part];
    };

    enum : uint32_t { IDLE = 0, BEGIN = 1, TRAV = 2, FINISH = 3 };
    uint32_t state = IDLE;
    uint32_t item = 0;
    ItemQueue iq(p, total_items);
    Path pt;
    pt.depth = 0;
    TriHit th;
    th.hit = false;
    th.t = 1e20f;
    th.u = th.v = 0.0f;
    th.slot = 0u;
    f3 inv = mk(0, 0, 0);
    int sp = 0;

    for (;;) {
        // ---- (1) hand items to idle lanes
        iq.refill(fresh_params(p), lane, total_items, S, tiles_x, sample_base, [&] { return state == IDLE; },
                  [&](uint32_t it, uint32_t x, uint32_t y, uint32_t sample_hash) {
                      start_path_hashed(fresh_params(p), x, y, y * width + x, sample_hash, pt);
                      item = it;
                      if (p.u.max_depth > 0u) {
                          state = BEGIN;
                      } else {
                          colors[it] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                          tl.paths++;
                      }
                  });
        if (__ballot(state != IDLE) == 0ull) {
            if (iq.drained()) break;
            continue;
        }

        // ---- (2) start of a segment: reset the traversal (shader.wgsl:283-307)
        if (state == BEGIN) {
            th.hit = false;
            th.t = 1e20f;
            th.u = th.v = 0.0f;
            th.slot = 0u;
            inv = mk(rcp_exact(pt.d.x), rcp_exact(pt.d.y), rcp_exact(pt.d.z));
            stack[0] = 0u;
            sp = 1;
            state = TRAV;
        }

        // ---- (3) node phase: pop until this lane has a leaf to test or its stack is empty
        uint32_t first = 0, count = 0;
        while (state == TRAV && count == 0u) {
            if (sp == 0) {
                state = FINISH;
                break;
            }
            sp--;
            const uint32_t node_idx = stack[sp * BLOCK];
            if (node_idx >= node_count) continue;
            const v4f n0 = node_q(node_idx, 0u), n1 = node_q(node_idx, 1u), n2f = node_q(node_idx, 2u);
            const uint32_t n_left = __float_as_uint(n2f.x), n_right = __float_as_uint(n2f.y),
                           n_first = __float_as_uint(n2f.z), n_count = __float_as_uint(n2f.w);
            if constexpr (STATS) tl.nodes++;
            if (!isect_aabb(pt.o, inv, mk(n0.x, n0.y, n0.z), mk(n1.x, n1.y, n1.z))) continue;
            if (n_count > 0u) {
                first = n_first;
                count = n_count;
            } else {
                if (n_left < node_count) {
                    stack[sp * BLOCK] = n_left;
                    sp++;
                }
                if (n_right < node_count) {
                    stack[sp * BLOCK] = n_right;
                    sp++;
                }
            }
        }

        // ---- (4) leaf phase: this lane's leaf (shader.wgsl:327-374), two triangles per step so
        // that six 16-byte loads are in flight per lane; candidates are offered in slot order
        {
            const uint32_t end = (first + count < p.index_len) ? first + count : p.index_len;  // guard :331
            uint32_t slot = first;
            for (; slot < end; slot++) {
                const v4f a = tri_q(slot, 0u), b = tri_q(slot, 1u), c = tri_q(slot, 2u);
                if (__float_as_uint(c.w) == 0u) continue;  // guard :336
                if constexpr (STATS) tl.tris++;
                const float before = th.t;
                test_slot(a, b, c, slot, pt.o, pt.d, th);
                if constexpr (STATS) tl.mesh_hits += (th.t != before) ? 1u : 0u;
            }
        }

        // ---- (5) traversal complete: ground, spheres, lights, shading, next ray
        if (state == FINISH) {
            const bool alive = segment_finish<STATS>(p, pt, th, stack, BLOCK, tl);
            if (alive) {
                state = BEGIN;
            } else {
                store_color(colors, item, pt.color);
                tl.paths++;
                state = IDLE;
            }
        }
    }
    flush_tally<STATS>(tl, p.counters);
}

// ======================================================= kernel: CHUNK ====
// The chunked walk (rb_internal.hpp, ChunkNode).  Two shapes of work alternate inside one
// persistent wavefront:
//   lane = RAY for the tree: every lane walks its own ray down the two-box nodes, nearer child first.  Nodes made from
//     the caller's tree carry the reference's boxes and are entered only if the reference's own slab test passes
//     (shader.wgsl:664-671, same operations) -- so a reference leaf is reached exactly when intersect_bvh reaches it --
//     and, like the library's own levels below the reference leaves, only if the ray enters the box, inflated by the
//     margin that bounds how far from its triangle the reference can report a hit, no later than the best t so far;
//   lane = TRIANGLE for the leaves: the (ray, chunk) pairs of all 64 lanes are pooled, and every round four of them
//     are tested by 16 lanes each -- one triangle per lane, records read as consecutive 16-byte pieces of three
//     arrays (a wavefront's load touches 4 x 256 B instead of 64 scattered records), the ray fetched from LDS, the
//     reference's intersect_triangle unchanged -- and a hit goes to its ray's best key with one LDS atomic min on
//     (t, rank in the reference's visit order): the reference's strict `t < closest` in visit order is exactly the
//     lexicographic minimum.
// A lane whose stack runs empty is shaded (segment_finish) and starts its next segment or a new path while the others
// keep walking, as in the other stepped kernels.
#ifndef RB_CHUNK_WAVES
#define RB_CHUNK_WAVES 5   // 96 registers, a few of them spilled outside the hot loops: + 3..7 % over 4 waves per SIMD (profiles/r03_chunk_steps.txt); 6 spills too much
#endif
#ifndef RB_CHUNK_NODE_LANES
#define RB_CHUNK_NODE_LANES 32   // keep stepping nodes while this many lanes are at one ...
#endif
#ifndef RB_CHUNK_NODE_STEPS
#define RB_CHUNK_NODE_STEPS 8    // ... but at most this many steps per outer iteration (r04: 8 instead of 5, + 1..2 %)
#endif
#ifndef RB_CHUNK_LEAF_LANES
#define RB_CHUNK_LEAF_LANES 8    // test chunks once this many lanes wait at one (or nobody is at a node)
#endif
#ifndef RB_CHUNK_FINISH_LANES
#define RB_CHUNK_FINISH_LANES 32 // shade once this many lanes have finished their walk (or nobody walks)
#endif
// Settled experiments, each measured in r03 (profiles/r03_chunk_steps.txt) and kept as tools/ablate/rb_forks.patch, not here:
// the margin as one box inflation instead of its two parts (C3 - 9 %), Sp by the largest component (- 3 %), the fixed c0
// instead of the ray's own cone bound (- 1..2 %), no chunks put aside (- 3..8 %), no prefetch of the next round (- 2 %).
// of the hit's error (21.4 |s| + 9.1 L) u L^2 / |a|, (11.2 |s| + 4.6 L) is how far the exact plane
// point Q* = o + t* d can be from the triangle's box -- that part inflates the box --, (10.2 |s| + 4.6 L) is |t^ - t*|, which
// only moves the hit along the ray.  4 u of kChunkKS are for the slab arithmetic done on the uninflated box (chunk_child).
// The stored bounds are of G / |a^| with G = |e1| |e2| where r02 had L^2 (E7's leading terms carry |e1| |e2|); the proviso of the
// bounds, 5.42 u L^2 / |a^| <= 0.05, is the host's business where it can be (rb_bvh.cpp pack_fac) and this test elsewhere.
constexpr float kChunkFMax = 1.5e5f;
constexpr float kChunkKS = 24.0f * 5.9604645e-8f * 1.01f;
constexpr float kChunkKP = 12.0f * 5.9604645e-8f * 1.01f;   // across
constexpr float kChunkKT = 11.0f * 5.9604645e-8f * 1.01f;   // along
constexpr float kChunkKD = 16.0f * 5.9604645e-8f * 1.01f;   // along, the relative part: 4 u t^ (t^ |d| <= |s| + 2.1 L), |d| = 1 +- 4 u, in units of Sp
constexpr uint32_t kChunkWaveLds = 64u * 32u + 64u * 8u + 128u * 4u;  // per wave: ray records, best keys, unit table
constexpr unsigned long long kChunkNoHit = 0x60AD78EC00000000ull;     // (bits of 1e20f) << 32: shader.wgsl:283-290
typedef __attribute__((address_space(3))) unsigned long long lds_u64;

// max(|a|, |b|) in one instruction (fmaxf(fabsf(a), fabsf(b)) compiles to three: each operand is quietened first)
DEV float max_abs(float a, float b) {
    float r;
    asm("v_max_f32 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// One child slot of a ChunkNode: enter it?  `order` = where the ray enters the box as stored (nearer child first).
// The slab values are the reference's (shader.wgsl:664-671 on the box as stored); the inflated box is derived from
// them per axis -- a box grown by mm enters mm |1 / d| earlier and leaves as much later -- so one set of operations
// serves the exact test and the conservative one.  NaN (0 * inf) always means "enter".
DEV bool chunk_child(v4f lo, v4f hi, uint32_t fac, v4f cone, bool exact, f3 o, f3 d, f3 inv, float best_t, float& order) {
    const f3 a = mk(lo.x, lo.y, lo.z) - o, b = mk(hi.x, hi.y, hi.z) - o;
    const f3 t0 = a * inv, t1 = b * inv;
    const float nx = fminf(t0.x, t1.x), ny = fminf(t0.y, t1.y), nz = fminf(t0.z, t1.z);
    const float fx = fmaxf(t0.x, t1.x), fy = fmaxf(t0.y, t1.y), fz = fmaxf(t0.z, t1.z);
    const float tmin = fmaxf(fmaxf(nx, ny), nz), tmax = fminf(fminf(fx, fy), fz);
    if (exact && !(tmax >= fmaxf(tmin, 0.0f))) return false;   // the reference does not enter this node
    // the bound of L^2 / |a^| this ray needs below the child
    // |a| = N |cos(d, n)| >= N lb for every triangle below, lb = the cone's lower bound of |cos| for THIS ray: the bound
    // (L^2 / N) / (0.95 lb) -- the stored one, made for |cos| >= c0, times c0 / lb -- up to the determinant floor,
    // which holds whatever the angle (all zeros = no cone: lb <= 0; tan = -1 = nothing below: the floor of nothing)
    const float lb = cone_cos_bound(d, cone);
    const float cap = __uint_as_float(fac & 0xFFFF0000u);
    const float fl = __uint_as_float(fac << 16) * (kFastGrazeCos * 1.00001f) * __builtin_amdgcn_rcpf(lb);
    const float f = (lb > 1e-6f && fl < cap) ? fl : cap;   // NaN -> cap
    // Sp >= |o - v0| + L / 2 for every triangle below (E7's L terms are less than half its |s| terms): farthest corner
    // (v_sqrt_f32 is within 1 ulp) + half the box's extents
    const float mx = max_abs(a.x, b.x), my = max_abs(a.y, b.y), mz = max_abs(a.z, b.z);
    const float sp_ = 1.001f * __builtin_amdgcn_sqrtf(__builtin_fmaf(mx, mx, __builtin_fmaf(my, my, mz * mz))) +
                      0.5f * (((b.x - a.x) + (b.y - a.y)) + (b.z - a.z));
    const float ix = fabsf(inv.x), iy = fabsf(inv.y), iz = fabsf(inv.z);
    // Q* = o + t* d, the exact plane point of an accepted hit, lies on the ray within mm of the triangle's box, so the ray's
    // line passes the box inflated by mm at parameters [tn, tf] that hold t*; what is reported, t^, is within dt of t*, has to
    // be positive and, for the winner, no larger than the best t so far
    const bool fin = f <= kChunkFMax;                                        // NaN -> always enter
    const float mm = fin ? sp_ * __builtin_fmaf(kChunkKP, f, kChunkKS) : 1e30f;
    const float dt = fin ? sp_ * __builtin_fmaf(kChunkKT, f, kChunkKD) : 1e30f;
    const float tn = fmaxf(fmaxf(__builtin_fmaf(-mm, ix, nx), __builtin_fmaf(-mm, iy, ny)), __builtin_fmaf(-mm, iz, nz));
    const float tf = fminf(fminf(__builtin_fmaf(mm, ix, fx), __builtin_fmaf(mm, iy, fy)), __builtin_fmaf(mm, iz, fz));
    // nearer child first by where the ray enters the box AS STORED, not the inflated one: a wide margin makes tn early for
    // every child that has one and says little about which child the ray meets first (r04: C3 - 17 % triangle tests, - 9 %
    // child tests, + 14 % segments/s; C5 + 6 %; any order is correct)
    order = tmin;
    return !(tf < tn) && !(tf < -dt) && !(tn - dt > best_t);
}

// cur is an internal node: descend into the nearer child that is entered, remember the other.  False when the walk is complete.
template <bool STATS>
DEV bool chunk_node_step(const KParams& p, uint32_t* stack, uint32_t stride, f3 o, f3 d, f3 inv, float best_t, uint32_t& cur,
                         int& sp, Tally<STATS>& tl) {
    const bool exact = (cur & kChunkExact) != 0u;
    const cf4p q = (cf4p)p.chunk_nodes + (size_t)(cur & 0x3FFFFFFFu) * 6u;
    const v4f l0 = q[0], l1 = q[1], r0 = q[2], r1 = q[3], lc = q[4], rc = q[5];
    const uint32_t lref = __float_as_uint(l0.w), rref = __float_as_uint(l1.w);
    float kl = 0.0f, kr = 0.0f;
    if constexpr (STATS) tl.nodes += (lref != kChunkNone ? 1u : 0u) + (rref != kChunkNone ? 1u : 0u);
    const bool vl = lref != kChunkNone && chunk_child(l0, l1, __float_as_uint(r0.w), lc, exact, o, d, inv, best_t, kl);
    const bool vr = rref != kChunkNone && chunk_child(r0, r1, __float_as_uint(r1.w), rc, exact, o, d, inv, best_t, kr);
    if (vl && vr) {
        // (the entry distance is not kept with the reference: dropping put-aside subtrees at the pop when a nearer hit
        // has turned up meanwhile was measured -- 8-byte stack entries -- and removes 0.3 % of the box tests)
        const bool left_first = !(kr < kl);
        stack[sp * stride] = left_first ? rref : lref;
        sp++;
        cur = left_first ? lref : rref;
        return true;
    }
    if (vl || vr) {
        cur = vl ? lref : rref;
        return true;
    }
    if (sp == 0) return false;
    sp--;
    cur = stack[sp * stride];
    return true;
}

// SPHTREE: the instantiation for scenes that also have a sphere tree (more than 64 spheres), whose per-lane walk runs inside
// segment_finish; the other one leaves that walk out of its register allocation
template <bool STATS, bool SPHTREE>
__global__ void __launch_bounds__(kTraceBlock, RB_CHUNK_WAVES) k_trace_chunk(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_stack[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t width = p.u.width;
    const uint32_t tiles_x = (width + 7u) / 8u;
    const uint32_t tiles_y = (p.local_rows + 7u) / 8u;
    const uint32_t S = p.n_passes * p.samples_per_pass;
    const uint32_t total_items = tiles_x * tiles_y * S * 64u;
    const uint32_t sample_base = p.first_pass * p.samples_per_pass;
    float4* __restrict__ colors = reinterpret_cast<float4*>(p.colors);
    uint32_t* const stack = stack_column(s_stack, tid);
    Tally<STATS> tl;
    // this wave's corner of LDS behind the traversal stacks
    unsigned char* const wl = reinterpret_cast<unsigned char*>(s_stack + p.stack_depth * kTraceBlock) + (tid >> 6) * kChunkWaveLds;
    lds_v4f* const rayrec = (lds_v4f*)wl;                    // [64][2]: {o, chunk reference}, {d, -}
    lds_u64* const best = (lds_u64*)(wl + 64u * 32u);        // [64]: (t bits) << 32 | rank
    lds_u32* const units = (lds_u32*)(wl + 64u * 40u);       // [128]: ray lane (| 64: its second chunk) of every pooled (ray, chunk) pair

    enum : uint32_t { IDLE = 0, BEGIN = 1, TRAV = 2, FINISH = 3 };
    uint32_t state = IDLE;
    uint32_t item = 0, cur = 0;
    uint32_t pend = kChunkNone;   // the chunk this lane has put aside (cur == kChunkNone: nothing else left to walk)
    ItemQueue iq(p, total_items);
    Path pt;
    pt.depth = 0;
    f3 inv = mk(0, 0, 0);
    unsigned long long key = kChunkNoHit;
    int sp = 0;
    // a lane whose walk arrives at a chunk while it holds none aside keeps the chunk for the next leaf
    // phase and goes on with the subtree it had put aside, so it takes part in twice as many node passes between two
    // waits (the best t it culls with is then one chunk behind: never wrong, rarely wasteful -- few chunk visits hit)
    auto set_aside = [&]() {
        if (state == TRAV && cur != kChunkNone && (cur & kChunkLeaf) != 0u && pend == kChunkNone) {
            pend = cur;
            if (sp == 0) {
                cur = kChunkNone;
            } else {
                sp--;
                cur = stack[sp * kTraceBlock];
            }
        }
    };

    for (;;) {
        // ---- (1) hand items to idle lanes
        iq.refill(fresh_params(p), lane, total_items, S, tiles_x, sample_base, [&] { return state == IDLE; },
                  [&](uint32_t it, uint32_t x, uint32_t y, uint32_t sample_hash) {
                      start_path_hashed(fresh_params(p), x, y, y * width + x, sample_hash, pt);
                      item = it;
                      if (p.u.max_depth > 0u) {
                          state = BEGIN;
                      } else {
                          colors[it] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                          tl.paths++;
                      }
                  });
        if (__ballot(state != IDLE) == 0ull) {
            if (iq.drained()) break;
            continue;
        }

        // ---- (2) start of a segment: the root's own box (shader.wgsl:283-315), then its two children
        if (state == BEGIN) {
            const KParams& fp = fresh_params(p);
            inv = mk(rcp_exact(pt.d.x), rcp_exact(pt.d.y), rcp_exact(pt.d.z));
            key = kChunkNoHit;
            sp = 0;
            const cf4p rn = (cf4p)fp.nodes;
            const v4f n0 = rn[0], n1 = rn[1];
            if constexpr (STATS) tl.nodes++;
            if (isect_aabb(pt.o, inv, mk(n0.x, n0.y, n0.z), mk(n1.x, n1.y, n1.z))) {
                cur = fp.chunk_root;
                state = TRAV;
            } else {
                state = FINISH;
            }
        }

        // ---- (3) tree: a few node steps while enough lanes are at a node
#pragma unroll 1
        for (int it = 0; it < RB_CHUNK_NODE_STEPS; ++it) {
            const bool at_node = state == TRAV && cur != kChunkNone && (cur & kChunkLeaf) == 0u;
            const uint32_t n = (uint32_t)__popcll(__ballot(at_node));
            if (n == 0u || (it > 0 && n < (uint32_t)RB_CHUNK_NODE_LANES)) break;
            if (at_node) {
                if (!chunk_node_step<STATS>(p, stack, kTraceBlock, pt.o, pt.d, inv, __uint_as_float((uint32_t)(key >> 32)), cur, sp, tl)) {
                    if (pend != kChunkNone) cur = kChunkNone;   // nothing left to walk, one chunk still to be tested
                    else
                    state = FINISH;
                }
                set_aside();
            }
        }

        // ---- (4) leaves: pool the (ray, chunk) pairs of the lanes that hold a chunk, 16 lanes per pair
        {
            const bool lf = state == TRAV && cur != kChunkNone && (cur & kChunkLeaf) != 0u;   // waits at a chunk
            const bool lp = state == TRAV && pend != kChunkNone;                              // holds one aside
            const unsigned long long m = __ballot(lf), mp = __ballot(lp);
            const uint32_t n_pend = (uint32_t)__popcll(mp), n_units = n_pend + (uint32_t)__popcll(m);
            const uint32_t n_node = (uint32_t)__popcll(__ballot(state == TRAV && cur != kChunkNone && !lf));
            if (n_units != 0u && (n_units >= (uint32_t)RB_CHUNK_LEAF_LANES || n_node == 0u)) {
                const unsigned long long below = (1ull << lane) - 1ull;
                if (lp) units[(uint32_t)__popcll(mp & below)] = lane;
                if (lf) units[n_pend + (uint32_t)__popcll(m & below)] = lane | 64u;
                if (lf || lp) {
                    const v4f r0 = {pt.o.x, pt.o.y, pt.o.z, __uint_as_float(pend)}, r1 = {pt.d.x, pt.d.y, pt.d.z, __uint_as_float(cur)};
                    rayrec[lane * 2u] = r0;
                    rayrec[lane * 2u + 1u] = r1;
                    best[lane] = key;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const cf4p ca = (cf4p)p.chunk_a, cb = (cf4p)p.chunk_b, cc = (cf4p)p.chunk_c;
                // one round = four pairs; the next round's ray records and triangle pieces are requested before
                // this round's are tested
                constexpr uint32_t kPairsPerRound = 64u / kChunkTris;
                struct Round {
                    v4f r0, r1, a, b, c;
                    uint32_t rl;
                    bool valid;
                };
                auto fetch = [&](uint32_t g0) {
                    Round r;
                    const uint32_t g = g0 + lane / kChunkTris;
                    const bool ok = g < n_units;
                    const uint32_t e = units[ok ? g : 0u];
                    r.rl = e & 63u;
                    r.r0 = rayrec[r.rl * 2u];
                    r.r1 = rayrec[r.rl * 2u + 1u];
                    const uint32_t ref = __float_as_uint((e & 64u) ? r.r1.w : r.r0.w), first = ref & 0x03FFFFFFu, cnt = ((ref >> 26) & 31u) + 1u;
                    const uint32_t j = lane & (kChunkTris - 1u);
                    r.valid = ok && j < cnt;
                    const uint32_t pos = first + (j < cnt ? j : 0u);
                    r.a = ca[pos];
                    r.b = cb[pos];
                    r.c = cc[pos];
                    return r;
                };
                Round nx = fetch(0u);
#pragma unroll 1
                for (uint32_t g0 = 0; g0 < n_units; g0 += kPairsPerRound) {
                    const Round r = nx;
                    if (g0 + kPairsPerRound < n_units) nx = fetch(g0 + kPairsPerRound);
                    if constexpr (STATS) tl.tris += r.valid ? 1u : 0u;
                    float u, v;
                    const float t = isect_triangle(mk(r.r0.x, r.r0.y, r.r0.z), mk(r.r1.x, r.r1.y, r.r1.z), mk(r.a.x, r.a.y, r.a.z),
                                                   mk(r.b.x, r.b.y, r.b.z), mk(r.c.x, r.c.y, r.c.z), u, v);
                    if (r.valid && t > 0.001f) {
                        const unsigned long long k = ((unsigned long long)__float_as_uint(t) << 32) | __float_as_uint(r.a.w);
                        __hip_atomic_fetch_min(&best[r.rl], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        if constexpr (STATS) tl.mesh_hits++;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (lf || lp) {
                    key = best[lane];
                    pend = kChunkNone;
                    if (lf || cur == kChunkNone) {   // the chunk the lane stood at is done, or there was nothing left to walk
                        if (sp == 0) {
                            state = FINISH;
                        } else {
                            sp--;
                            cur = stack[sp * kTraceBlock];
                        }
                    }
                    set_aside();
                }
            }
        }

        // ---- (5) finished walks: the winner's record, the rest of the segment (shading), next ray
        {
            const uint32_t n_fin = (uint32_t)__popcll(__ballot(state == FINISH));
            const uint32_t n_trav = (uint32_t)__popcll(__ballot(state == TRAV));
            if (n_fin != 0u && (n_fin >= (uint32_t)RB_CHUNK_FINISH_LANES || n_trav == 0u) && state == FINISH) {
                TriHit th;
                th.hit = key != kChunkNoHit;
                th.t = 1e20f;
                th.u = th.v = 0.0f;
                th.slot = 0u;
                if (th.hit) {
                    // (t, u, v) of the winner again from its prepared record: the same operations on the same values
                    const KParams& fp = fresh_params(p);
                    th.slot = cptr(fp.chunk_rank_slot)[(uint32_t)key];
                    const cf4p tp = (cf4p)fp.ptris + (size_t)th.slot * 4u;
                    const v4f a = tp[0], b = tp[1], c = tp[2];
                    th.t = isect_triangle(pt.o, pt.d, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), th.u, th.v);
                }
                const bool alive = segment_finish<STATS, SPHTREE>(p, pt, th, stack, kTraceBlock, tl);
                if (alive) {
                    state = BEGIN;
                } else {
                    store_color(colors, item, pt.color);
                    tl.paths++;
                    state = IDLE;
                }
            }
        }
    }
    flush_tally<STATS>(tl, p.counters);
}

// RB_FLAG_FAST_BVH, stepped: the opt-in walk (FastWalk, rb_device_intersect.hpp) with the same
// scheduling idea as k_trace_bvh.  Per-ray walk lengths differ a lot (a few dozen dependent node
// fetches), so in the per-segment form (k_trace) a wavefront waits for its slowest lane with ~20 %
// of its lanes busy.  Here every lane keeps its walk state in registers; the wavefront issues node
// steps and leaf steps separately, each when enough lanes need one, and as soon as fewer than
// RB_FAST_KEEP lanes are still walking, the finished ones are shaded (segment_finish) and continue
// with their next segment or a new path from the queue.
#ifndef RB_FAST_KEEP
#define RB_FAST_KEEP 32
#endif
#ifndef RB_FAST_NODE_STEPS
#define RB_FAST_NODE_STEPS 2
#endif
#ifndef RB_FAST_WAVES
#define RB_FAST_WAVES 4
#endif
// What trace_stepped needs from a resumable walk: the opt-in triangle walk ...
template <bool STATS>
struct TriangleWalkPolicy {
    static constexpr int kNodeSteps = RB_FAST_NODE_STEPS;
    FastWalk<STATS> w;
    DEV uint32_t kind() const { return w.kind(); }
    DEV bool step(const KParams& p, uint32_t* stack, Tally<STATS>& tl) { return w.step(p, stack, kTraceBlock, tl); }
    DEV void init(const KParams& p) { w.begin(p, mk(0, 0, 0), mk(0, 0, 1)); }
    DEV void begin(const KParams& p, const Path& pt, uint32_t*, Tally<STATS>&) { w.begin(p, pt.o, pt.d); }
    DEV bool finish(const KParams& p, Path& pt, uint32_t* stack, Tally<STATS>& tl) {
        return segment_finish<STATS>(p, pt, w.h, stack, kTraceBlock, tl);
    }
};
template <bool STATS, class Walk>
DEV void trace_stepped(const KParams& p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_stack[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t width = p.u.width;
    const uint32_t tiles_x = (width + 7u) / 8u;
    const uint32_t tiles_y = (p.local_rows + 7u) / 8u;
    const uint32_t S = p.n_passes * p.samples_per_pass;
    const uint32_t total_items = tiles_x * tiles_y * S * 64u;
    const uint32_t sample_base = p.first_pass * p.samples_per_pass;
    float4* __restrict__ colors = reinterpret_cast<float4*>(p.colors);
    uint32_t* const stack = stack_column(s_stack, tid);
    Tally<STATS> tl;

    enum : uint32_t { IDLE = 0, BEGIN = 1, TRAV = 2, FINISH = 3 };
    uint32_t state = IDLE;
    uint32_t item = 0;
    ItemQueue iq(p, total_items);
    Path pt;
    pt.depth = 0;
    Walk w;
    w.init(p);

    for (;;) {
        // ---- (1) hand items to idle lanes
        iq.refill(fresh_params(p), lane, total_items, S, tiles_x, sample_base, [&] { return state == IDLE; },
                  [&](uint32_t it, uint32_t x, uint32_t y, uint32_t sample_hash) {
                      start_path_hashed(fresh_params(p), x, y, y * width + x, sample_hash, pt);
                      item = it;
                      if (p.u.max_depth > 0u) {
                          state = BEGIN;
                      } else {
                          colors[it] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                          tl.paths++;
                      }
                  });
        if (__ballot(state != IDLE) == 0ull) {
            if (iq.drained()) break;
            continue;
        }

        // ---- (2) start of a segment
        if (state == BEGIN) {
            w.begin(p, pt, stack, tl);
            state = TRAV;
        }

        // ---- (3) walk: four kinds of step (FastWalk: node / leaf of the library's tree, then node / leaf chunk of the second
        // pass over the reference tree); every pass runs the kind most lanes are waiting for, node kinds a few steps per
        // vote (a lane reaches a leaf only every few nodes), then on while enough lanes are still walking
        for (;;) {
            const uint32_t k = state == TRAV ? w.kind() : 4u;
            const uint32_t n0 = (uint32_t)__popcll(__ballot(k == 0u)), n1 = (uint32_t)__popcll(__ballot(k == 1u)),
                           n2 = (uint32_t)__popcll(__ballot(k == 2u)), n3 = (uint32_t)__popcll(__ballot(k == 3u));
            uint32_t pick = 0u, best = n0;
            if (n1 > best) { pick = 1u; best = n1; }
            if (n2 > best) { pick = 2u; best = n2; }
            if (n3 > best) { pick = 3u; best = n3; }
            if (best == 0u) break;
            if (k == pick) {
                if (!w.step(p, stack, tl)) state = FINISH;
                if ((pick & 1u) == 0u)
                    for (int extra = 1; extra < Walk::kNodeSteps; ++extra)
                        if (state == TRAV && w.kind() == pick && !w.step(p, stack, tl)) state = FINISH;
            }
            if ((uint32_t)__popcll(__ballot(state == TRAV)) < (uint32_t)RB_FAST_KEEP) break;
        }

        // ---- (4) finished walks: the rest of the segment (shading), next ray
        if (state == FINISH) {
            const bool alive = w.finish(p, pt, stack, tl);
            if (alive) {
                state = BEGIN;
            } else {
                store_color(colors, item, pt.color);
                tl.paths++;
                state = IDLE;
            }
        }
    }
    flush_tally<STATS>(tl, p.counters);
}

template <bool STATS>
__global__ void __launch_bounds__(kTraceBlock, RB_FAST_WAVES) k_trace_fast(const KParams p) {
    trace_stepped<STATS, TriangleWalkPolicy<STATS>>(p);
}

// ===================================================== kernel: SPHERES ====
// Scenes with more than 64 spheres and no multi-node triangle tree (BASELINE C4: 10^6 spheres).  The reference scans every
// sphere on every segment (shader.wgsl:574-586); the library walks its own tree (rb_internal.hpp SphereNode; built on the
// device, rb_build.hip) between segment_pre (ground, the at most single-node triangle list) and segment_post (lights,
// shading), with the two shapes of work of k_trace_chunk:
//   lane = RAY for the tree: every lane walks its own ray down the 4-wide nodes (half as many dependent fetches as with
//     two-box nodes: the walk waits on them), a child skipped only when no sphere below it can be REPORTED hit nearer than
//     the best t (sphere_child / sphere_node_step, rb_device_shade.hpp);
//   lane = SPHERE for the leaves: the (ray, leaf) pairs of all 64 lanes are pooled, every round 64 / kSphLeaf of them are
//     tested by kSphLeaf lanes each -- one sphere per lane, the leaf's 16-byte {centre, radius} records read as consecutive
//     bytes, the ray from LDS -- in two steps: the reference's discriminant for every lane (same operations, same bits:
//     `disc < 0` is the reference's own early return), and only the survivors, compacted over the rounds into one list, go
//     through the whole intersect_sphere (sqrt, divisions), 64 at a time.  A hit goes to its ray's best key with one LDS
//     atomic min on (t bits) << 32 | original index: the key starts at (closest t of the earlier categories) << 32, so an
//     equal t never displaces the ground or a triangle, and among spheres the lexicographic minimum is the linear scan's
//     winner (strict `t < closest` in index order).
// r03's form (every lane its own node and its own <= 4-sphere leaf: 8 L1 accesses per sphere test, 46 % of the lanes busy,
// 42 % of the wave cycles waiting) made 4.48 G segments/s on C4.
#ifndef RB_SPH_WAVES
#define RB_SPH_WAVES 4   // 128 registers: nothing spilled; the LDS (stack columns of 3 entries per level + 3.75 KiB per wave) holds four blocks per CU anyway
#endif
#ifndef RB_SPH_NODE_LANES
#define RB_SPH_NODE_LANES 24
#endif
#ifndef RB_SPH_NODE_STEPS
#define RB_SPH_NODE_STEPS 8
#endif
#ifndef RB_SPH_LEAF_LANES
#define RB_SPH_LEAF_LANES 8
#endif
#ifndef RB_SPH_FINISH_LANES
#define RB_SPH_FINISH_LANES 48
#endif
#ifndef RB_SPH_PER_LANE
#define RB_SPH_PER_LANE 2   // spheres a lane tests per round: a leaf's kSphLeaf spheres go to kSphLeaf / 2 lanes, which fetch the ray once for two tests
#endif
constexpr uint32_t kSphPerLane = RB_SPH_PER_LANE;
static_assert(kSphPerLane == 1 || kSphPerLane == 2, "one or two spheres per lane and round (four: 36 spilled registers, - 13 %)");
constexpr uint32_t kSphWaveLds = 64u * 32u + 64u * 8u + 64u * 4u + 128u * 4u + 128u * 4u;   // per wave: ray records, best keys, a = d.d, units, survivors

template <bool STATS>
__global__ void __launch_bounds__(kTraceBlock, RB_SPH_WAVES) k_trace_sph(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_stack[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t width = p.u.width;
    const uint32_t tiles_x = (width + 7u) / 8u;
    const uint32_t tiles_y = (p.local_rows + 7u) / 8u;
    const uint32_t S = p.n_passes * p.samples_per_pass;
    const uint32_t total_items = tiles_x * tiles_y * S * 64u;
    const uint32_t sample_base = p.first_pass * p.samples_per_pass;
    float4* __restrict__ colors = reinterpret_cast<float4*>(p.colors);
    uint32_t* const stack = stack_column(s_stack, tid);
    Tally<STATS> tl;
    // this wave's corner of LDS behind the traversal stacks
    unsigned char* const wl = reinterpret_cast<unsigned char*>(s_stack + p.stack_depth * kTraceBlock) + (tid >> 6) * kSphWaveLds;
    lds_v4f* const rayrec = (lds_v4f*)wl;                           // [64][2]: {o, leaf put aside}, {d, leaf stood at}
    lds_u64* const best = (lds_u64*)(wl + 64u * 32u);               // [64]: (t bits) << 32 | sphere index
    typedef __attribute__((address_space(3))) float lds_f32;
    lds_f32* const ray_a = (lds_f32*)(wl + 64u * 40u);              // [64]: d . d
    lds_u32* const units = (lds_u32*)(wl + 64u * 44u);              // [128]: ray lane (| 64: the leaf it stands at) of every pooled (ray, leaf) pair
    lds_u32* const cands = (lds_u32*)(wl + 64u * 44u + 128u * 4u);  // [128]: pair << 4 | sphere of the pair's leaf: discriminant >= 0

    enum : uint32_t { IDLE = 0, BEGIN = 1, TRAV = 2, FINISH = 3 };
    uint32_t state = IDLE;
    uint32_t item = 0, cur = kSphNone, pend = kSphNone;
    ItemQueue iq(p, total_items);
    Path pt;
    pt.depth = 0;
    TriHit th;
    th.hit = false;
    th.t = 1e20f;
    th.u = th.v = 0.0f;
    th.slot = 0u;
    SegState st;
    st.closest_t = 1e20f;
    st.kind = K_NONE;
    st.uvx = st.uvy = 0.0f;
    st.use_tex = st.tri_won_a = false;
    f3 inv = mk(0, 0, 0);
    float aa = 1.0f;
    uint32_t dneg = 0u;
    unsigned long long key = 0ull;
    int sp = 0;
    // a lane whose walk arrives at a leaf while it holds none aside keeps the leaf for the next leaf phase and goes on with
    // the subtree it had put aside (its best t is then one leaf behind: never wrong)
    auto set_aside = [&]() {
        if (state == TRAV && cur != kSphNone && (cur & 0x80000000u) != 0u && pend == kSphNone) {
            pend = cur;
            if (sp == 0) {
                cur = kSphNone;
            } else {
                sp--;
                cur = stack[sp * kTraceBlock];
            }
        }
    };

    for (;;) {
        // ---- (1) hand items to idle lanes
        iq.refill(fresh_params(p), lane, total_items, S, tiles_x, sample_base, [&] { return state == IDLE; },
                  [&](uint32_t it, uint32_t x, uint32_t y, uint32_t sample_hash) {
                      start_path_hashed(fresh_params(p), x, y, y * width + x, sample_hash, pt);
                      item = it;
                      if (p.u.max_depth > 0u) {
                          state = BEGIN;
                      } else {
                          colors[it] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                          tl.paths++;
                      }
                  });
        if (__ballot(state != IDLE) == 0ull) {
            if (iq.drained()) break;
            continue;
        }

        // ---- (2) start of a segment: ground and the triangle list (shader.wgsl:552-571), then the root of the sphere tree
        if (state == BEGIN) {
            const KParams& fp = fresh_params(p);
            th = intersect_bvh<STATS, false>(fp, pt.o, pt.d, stack, kTraceBlock, tl);
            st = segment_pre<STATS>(fp, pt, th, tl);
            // (1 / d only steers the walk: the hardware's reciprocal, within 1 ulp; the margin of sphere_child allows for it)
            inv = mk(__builtin_amdgcn_rcpf(pt.d.x), __builtin_amdgcn_rcpf(pt.d.y), __builtin_amdgcn_rcpf(pt.d.z));
            aa = dot(pt.d, pt.d);
            dneg = sph_dir_signs(pt.d);
            key = (unsigned long long)__float_as_uint(st.closest_t) << 32;
            cur = fp.sph_root;
            pend = kSphNone;
            sp = 0;
            state = TRAV;
            set_aside();
        }

        // ---- (3) tree: a few node steps while enough lanes are at a node
#pragma unroll 1
        for (int it = 0; it < RB_SPH_NODE_STEPS; ++it) {
            const bool at_node = state == TRAV && cur != kSphNone && (cur & 0x80000000u) == 0u;
            const uint32_t n = (uint32_t)__popcll(__ballot(at_node));
            if (n == 0u || (it > 0 && n < (uint32_t)RB_SPH_NODE_LANES)) break;
            if (at_node) {
                const uint32_t nxt = sphere_node_step(p, cur, pt.o, inv, dneg, __uint_as_float((uint32_t)(key >> 32)), [&](uint32_t ref) {
                    stack[sp * kTraceBlock] = ref;
                    sp++;
                });
                if (nxt != kSphNone) {
                    cur = nxt;
                } else if (sp != 0) {
                    sp--;
                    cur = stack[sp * kTraceBlock];
                } else if (pend != kSphNone) {
                    cur = kSphNone;   // nothing left to walk, one leaf still to be tested
                } else {
                    state = FINISH;
                }
                set_aside();
            }
        }

        // ---- (4) leaves: pool the (ray, leaf) pairs of the lanes that hold a leaf, kSphLeaf lanes per pair
        {
            const bool lf = state == TRAV && cur != kSphNone && (cur & 0x80000000u) != 0u;   // stands at a leaf
            const bool lp = state == TRAV && pend != kSphNone;                               // holds one aside
            const unsigned long long m = __ballot(lf), mp = __ballot(lp);
            const uint32_t n_pend = (uint32_t)__popcll(mp), n_units = n_pend + (uint32_t)__popcll(m);
            const uint32_t n_node = (uint32_t)__popcll(__ballot(state == TRAV && cur != kSphNone && !lf));
            if (n_units != 0u && (n_units >= (uint32_t)RB_SPH_LEAF_LANES || n_node == 0u)) {
                const unsigned long long below = (1ull << lane) - 1ull;
                if (lp) units[(uint32_t)__popcll(mp & below)] = lane;
                if (lf) units[n_pend + (uint32_t)__popcll(m & below)] = lane | 64u;
                if (lf || lp) {
                    const v4f r0 = {pt.o.x, pt.o.y, pt.o.z, __uint_as_float(pend)}, r1 = {pt.d.x, pt.d.y, pt.d.z, __uint_as_float(cur)};
                    rayrec[lane * 2u] = r0;
                    rayrec[lane * 2u + 1u] = r1;
                    best[lane] = key;
                    ray_a[lane] = aa;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const cf4p leafs = (cf4p)p.sph_leaf;
                const RB_CONST uint32_t* ids = cptr(p.sph_id);
                constexpr uint32_t kLanesPerUnit = kSphLeaf / kSphPerLane, kUnitsPerRound = 64u / kLanesPerUnit;
                // the survivors of the discriminant, 64 at a time (or what is left): the whole intersect_sphere
                uint32_t n_cand = 0u;
                auto flush = [&](uint32_t k) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (lane < k) {
                        const uint32_t c = cands[n_cand - k + lane], g = c >> 4, e = units[g], rl = e & 63u;
                        const v4f r0 = rayrec[rl * 2u], r1 = rayrec[rl * 2u + 1u];
                        const uint32_t pos = sph_leaf_first(__float_as_uint((e & 64u) ? r1.w : r0.w)) + (c & 15u);
                        const v4f cr = leafs[pos];
                        const uint32_t id = ids[pos];
                        const float t = isect_sphere(mk(r0.x, r0.y, r0.z), mk(r1.x, r1.y, r1.z), ray_a[rl], mk(cr.x, cr.y, cr.z), cr.w);
                        if (t > 0.001f) {
                            const unsigned long long kk = ((unsigned long long)__float_as_uint(t) << 32) | id;
                            __hip_atomic_fetch_min(&best[rl], kk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        }
                    }
                    n_cand -= k;
                };
                // one round = kUnitsPerRound pairs; a pair's leaf is tested by kLanesPerUnit lanes, kSphPerLane spheres each; the
                // next round's ray records and sphere records are requested before this round's are tested
                struct Round {
                    v4f r0, r1, cr[kSphPerLane];
                    float a;
                    uint32_t tag;     // pair << 4 | first sphere of this lane
                    uint32_t cnt;     // spheres in the leaf (0: no pair for this lane)
                };
                auto fetch = [&](uint32_t g0) {
                    Round r;
                    const uint32_t g = g0 + lane / kLanesPerUnit;
                    const bool ok = g < n_units;
                    const uint32_t e = units[ok ? g : 0u], rl = e & 63u;
                    r.r0 = rayrec[rl * 2u];
                    r.r1 = rayrec[rl * 2u + 1u];
                    r.a = ray_a[rl];
                    const uint32_t ref = __float_as_uint((e & 64u) ? r.r1.w : r.r0.w), j = lane & (kLanesPerUnit - 1u);
                    r.cnt = ok ? sph_leaf_count(ref) : 0u;
#pragma unroll
                    for (uint32_t k = 0; k < kSphPerLane; k++) {
                        const uint32_t jj = j + k * kLanesPerUnit;
                        r.cr[k] = leafs[sph_leaf_first(ref) + (jj < r.cnt ? jj : 0u)];
                    }
                    r.tag = (g << 4) | j;
                    return r;
                };
                Round nx = fetch(0u);
#pragma unroll 1
                for (uint32_t g0 = 0; g0 < n_units; g0 += kUnitsPerRound) {
                    const Round r = nx;
                    if (g0 + kUnitsPerRound < n_units) nx = fetch(g0 + kUnitsPerRound);
                    const f3 ro = mk(r.r0.x, r.r0.y, r.r0.z), d = mk(r.r1.x, r.r1.y, r.r1.z);
#pragma unroll
                    for (uint32_t k = 0; k < kSphPerLane; k++) {
                        const bool valid = (r.tag & 15u) + k * kLanesPerUnit < r.cnt;
                        if constexpr (STATS) tl.spheres += valid ? 1u : 0u;
                        // shader.wgsl:194-199, the operations of isect_sphere up to its first return
                        const f3 oc = ro - mk(r.cr[k].x, r.cr[k].y, r.cr[k].z);
                        const float half_b = dot(oc, d);
                        const float c = dot(oc, oc) - r.cr[k].w * r.cr[k].w;
                        const float disc = half_b * half_b - r.a * c;
                        const bool cand = valid && !(disc < 0.0f);
                        const unsigned long long cm = __ballot(cand);
                        if (cm != 0ull) {
                            if (cand) cands[n_cand + (uint32_t)__popcll(cm & below)] = r.tag + k * kLanesPerUnit;
                            n_cand += (uint32_t)__popcll(cm);
                            if (n_cand >= 64u) flush(64u);
                        }
                    }
                }
                if (n_cand != 0u) flush(n_cand);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (lf || lp) {
                    key = best[lane];
                    pend = kSphNone;
                    if (lf || cur == kSphNone) {   // the leaf the lane stood at is done, or there was nothing left to walk
                        if (sp == 0) {
                            state = FINISH;
                        } else {
                            sp--;
                            cur = stack[sp * kTraceBlock];
                        }
                    }
                    set_aside();
                }
            }
        }

        // ---- (5) finished walks: lights, the winner's record, shading (shader.wgsl:590-660), next ray
        {
            const uint32_t n_fin = (uint32_t)__popcll(__ballot(state == FINISH));
            const uint32_t n_trav = (uint32_t)__popcll(__ballot(state == TRAV));
            if (n_fin != 0u && (n_fin >= (uint32_t)RB_SPH_FINISH_LANES || n_trav == 0u) && state == FINISH) {
                float closest_t = st.closest_t;
                uint32_t sphere_idx = 0xFFFFFFFFu;
                if (key < ((unsigned long long)__float_as_uint(st.closest_t) << 32)) {   // a sphere strictly nearer than the earlier categories
                    closest_t = __uint_as_float((uint32_t)(key >> 32));
                    sphere_idx = (uint32_t)key;
                }
                const bool alive = segment_post<STATS>(fresh_params(p), pt, th, st, closest_t, sphere_idx, tl);
                if (alive) {
                    state = BEGIN;
                } else {
                    store_color(colors, item, pt.color);
                    tl.paths++;
                    state = IDLE;
                }
            }
        }
    }
    flush_tally<STATS>(tl, p.counters);
}

// Phase 2: ordered accumulation + tone map + pack.  One wavefront per 8x8 tile,
// lane = pixel; each sample row is a contiguous 1 KiB read.
__global__ void __launch_bounds__(256) k_accumulate(const KParams p) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t width = p.u.width;
    const uint32_t tiles_x = (width + 7u) / 8u;
    const uint32_t tiles_y = (p.local_rows + 7u) / 8u;
    if (tile >= tiles_x * tiles_y) return;
    const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const uint32_t x = tx * 8u + (lane & 7u), ly = ty * 8u + (lane >> 3);
    if (x >= width || ly >= p.local_rows) return;
    if (global_row(p, ly) >= p.u.height) return;
    const uint32_t S = p.n_passes * p.samples_per_pass;
    const float4 a4 = reinterpret_cast<const float4*>(p.accum_in)[(size_t)ly * width + x];
    f3 acc = mk(a4.x, a4.y, a4.z);
    uint32_t total = f2u(a4.w);
    const nt_f4* __restrict__ c = reinterpret_cast<const nt_f4*>(p.colors) + ((size_t)tile * S) * 64u + lane;
    uint32_t s = 0;
    for (; s + 8u <= S; s += 8u) {
        nt_f4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = __builtin_nontemporal_load(&c[(size_t)(s + k) * 64u]);
#pragma unroll
        for (int k = 0; k < 8; k++) acc = acc + mk(v[k].x, v[k].y, v[k].z);
    }
    for (; s < S; s++) {
        const nt_f4 v = __builtin_nontemporal_load(&c[(size_t)s * 64u]);
        acc = acc + mk(v.x, v.y, v.z);
    }
    total += S;
    store_pixel(p, x, ly, acc, total);
}

// ============================================================ prep kernel ==
// Gathers triangles into bvh_indices order and hoists the per-triangle
// invariants (edge1, edge2, geometric normal) of shader.wgsl:249-250,351.
__global__ void k_prep_tris(const rb_gpu_triangle* tris, uint32_t tri_count, const uint32_t* indices,
                            uint32_t index_len, PrepTri* out, PrepTriShade* shade) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= index_len) return;
    PrepTri t;
    PrepTriShade s;
    const uint32_t id = indices[slot];
    if (id >= tri_count) {  // shader.wgsl:336 `continue`
        t = PrepTri{};
        s = PrepTriShade{};
        t.valid = 0u;
    } else {
        const rb_gpu_triangle g = tris[id];
        const f3 v0 = ld3(g.v0), v1 = ld3(g.v1), v2 = ld3(g.v2);
        const f3 e1 = v1 - v0, e2 = v2 - v0;
        const f3 n = normalize(cross(e1, e2));
        t.v0[0] = v0.x; t.v0[1] = v0.y; t.v0[2] = v0.z;
        t.e1[0] = e1.x; t.e1[1] = e1.y; t.e1[2] = e1.z;
        t.e2[0] = e2.x; t.e2[1] = e2.y; t.e2[2] = e2.z;
        t.n[0] = n.x; t.n[1] = n.y; t.n[2] = n.z;
        t.tri_id = id;
        t.mesh_index = g.mesh_index;
        t.valid = 1u;
        t._pad = 0u;
        s.v0_index = g.v0_index;
        s.v1_index = g.v1_index;
        s.v2_index = g.v2_index;
        s._pad = 0u;
    }
    out[slot] = t;
    shade[slot] = s;
}

// Per-material invariants of the shading branch (shader.wgsl:615-623,637), written into the pad
// words of the DEVICE copy of each Material (the caller's buffers are never touched).
__global__ void k_prep_materials(unsigned char* first_material, uint32_t stride, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    rb_material* m = reinterpret_cast<rb_material*>(first_material + (size_t)i * stride);
    const float specular_strength = (m->specular[0] + m->specular[1] + m->specular[2]) / 3.0f;
    const float diffuse_strength = (m->diffuse[0] + m->diffuse[1] + m->diffuse[2]) / 3.0f;
    const bool is_metal = specular_strength > 0.01f && diffuse_strength < 0.01f;
    const float fuzz = fminf(fmaxf(1.0f - (m->shininess / 1000.0f), 0.0f), 1.0f);
    m->_pad1 = fuzz;
    m->_pad2 = is_metal ? 1u : 0u;
}

// Copies prepared triangles into the fast tree's leaf order.
__global__ void k_gather_tris(const PrepTri* ptris, const uint32_t* slots, uint32_t n, PrepTri* out) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) out[j] = ptris[slots[j]];
}

// Prepared triangles into the chunked walk's order, as three arrays of 16-byte pieces so that 16 lanes testing the 16
// triangles of a chunk read 256 consecutive bytes per piece.
__global__ void k_chunk_gather(const PrepTri* ptris, const uint32_t* pos_slot, const uint32_t* pos_rank, uint32_t n, float4* a,
                               float4* b, float4* c) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const PrepTri t = ptris[pos_slot[j]];
    a[j] = make_float4(t.v0[0], t.v0[1], t.v0[2], __uint_as_float(pos_rank[j]));
    b[j] = make_float4(t.e1[0], t.e1[1], t.e1[2], 0.0f);
    c[j] = make_float4(t.e2[0], t.e2[1], t.e2[2], 0.0f);
}

// Multi-GPU assembly on the root device: every rank's padded stripe buffer arrives
// back to back in `gathered`; stripe s of the frame belongs to rank s % n, which stores its stripes in order.
__global__ void __launch_bounds__(256) k_deinterleave(const uint32_t* __restrict__ gathered, uint32_t* __restrict__ frame,
                                                       uint32_t width, uint32_t height, uint32_t padded_rows,
                                                       uint32_t stripe_rows, uint32_t n) {
    const uint32_t x = blockIdx.x * 256u + threadIdx.x;
    if (x >= width) return;
    for (uint32_t y = blockIdx.y; y < height; y += gridDim.y) {   // (a grid's y extent stops at 65 535: taller frames loop)
        const uint32_t s = y / stripe_rows, r = y - s * stripe_rows;
        const uint32_t rank = s % n, local_row = (s / n) * stripe_rows + r;
        frame[(size_t)y * width + x] = gathered[((size_t)rank * padded_rows + local_row) * width + x];
    }
}

// Exhaustive check of rcp_newton against the compiler's correctly rounded 1/b: every one of the
// 2^23 significands at biased exponent `expo`, both signs.  mismatch[0] counts differing results
// among rcp_safe inputs; mismatch[1..] records up to 15 offending bit patterns.
__global__ void k_rcp_exhaustive(uint32_t expo, uint32_t* mismatch) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= (1u << 23)) return;
    for (uint32_t sign = 0; sign < 2; sign++) {
        const uint32_t bits = (sign << 31) | ((expo & 0xFFu) << 23) | m;
        const float b = __uint_as_float(bits);
        float want, got;
        if (expo & 0x100u) {  // mode 1: sqrt (positive operands only)
            if (sign) continue;
            want = sqrtf(b);
            got = sqrt_exact(b);
        } else {
            if (!rcp_safe(b)) continue;
            want = 1.0f / b;
            got = rcp_newton(b);
        }
        if (__float_as_uint(want) != __float_as_uint(got)) {
            const uint32_t k = atomicAdd(&mismatch[0], 1u);
            if (k < 15u) mismatch[1u + k] = bits;
        }
    }
}

// Exhaustive check of div_newton: thread = one denominator significand (biased exponent eb),
// loop over `a_count` numerator significands starting at a_begin (biased exponent ea).
__global__ void k_div_exhaustive(uint32_t b_begin, uint32_t ea, uint32_t eb, uint32_t a_begin, uint32_t a_count,
                                 unsigned long long* mismatch) {
    const uint32_t mb = b_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (mb >= (1u << 23)) return;
    const float b = __uint_as_float((eb << 23) | mb);
    if (!div_safe_den(b)) return;
    const float y = rcp_newton(b);
    unsigned long long bad = 0;
    uint32_t first_bad = 0;
    for (uint32_t i = 0; i < a_count; i++) {
        const float a = __uint_as_float((ea << 23) | ((a_begin + i) & 0x7FFFFFu));
        const float want = a / b;
        const float got = div_newton(a, b, y);
        if (__float_as_uint(want) != __float_as_uint(got)) {
            if (bad == 0) first_bad = __float_as_uint(a);
            bad++;
        }
    }
    if (bad) {
        const unsigned long long k = atomicAdd(&mismatch[0], bad);
        if (k < 7ull) {
            mismatch[1 + 2 * k] = first_bad;
            mismatch[2 + 2 * k] = __float_as_uint(b);
        }
    }
}

// Ceiling of the L1 / texture-address path, measured where the bench runs: every lane gathers 16 bytes from its own
// pseudo-random 128-byte line of a table that fits in L2 (the access pattern of a lane-per-ray tree walk), eight
// or sixteen independent loads in flight per lane.  A wave instruction then costs 64 L1 accesses (rocprofv3 counts
// TCP_TOTAL_CACHE_ACCESSES = 1.000 per lane load on it: profiles/r03_l1_ceiling.txt); accesses per second = the number
// the mesh kernels' TCP_TOTAL_CACHE_ACCESSES rate is compared with (bench.py, roofline.l1).
template <int INFLIGHT>
__global__ void __launch_bounds__(256) k_l1_gather(const float4* __restrict__ table, uint32_t lines_mask, uint32_t rounds, float4* sink) {
    // per-lane line sequence: an odd stride through the table (every line visited, no two lanes of a wave on one line
    // while the table has >= 64 lines), piece = lane & 7; one multiply-add per load keeps the address arithmetic out of the way
    const uint32_t lane_id = blockIdx.x * 256u + threadIdx.x;
    uint32_t line = lane_id * 2654435761u;
    const uint32_t step = (lane_id * 40503u) | 1u;
    const uint32_t piece = lane_id & 7u;
    float4 acc = make_float4(0, 0, 0, 0);
    for (uint32_t r = 0; r < rounds; r++) {
        float4 v[INFLIGHT];
#pragma unroll
        for (int k = 0; k < INFLIGHT; k++) {
            line += step;
            v[k] = table[(size_t)(line & lines_mask) * 8u + piece];
        }
#pragma unroll
        for (int k = 0; k < INFLIGHT; k++) acc.x += v[k].x, acc.y += v[k].y, acc.z += v[k].z, acc.w += v[k].w;
    }
    if (acc.x == 12345.678f) sink[0] = acc;   // never true for the zero-filled table: keeps the loads alive
}

// Exposes the device's /, sqrt, normalize and u32->f32 to the parity tests.
__global__ void k_debug_math(const float* a, const float* b, float* out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = a[i], y = b[i];
    out[i] = x / y;
    out[n + i] = sqrtf(fabsf(x));
    const f3 v = normalize(mk(x, y, x * y));
    out[2 * n + i] = v.x;
    out[3 * n + i] = v.y;
    out[4 * n + i] = v.z;
    out[5 * n + i] = (float)__float_as_uint(x) / 4294967296.0f;
    out[6 * n + i] = fminf(fmaxf(x, y), x * 0.0f);
    out[7 * n + i] = dot(mk(x, y, x), mk(y, y, x));
}

}  // namespace

// ================================================================ launchers ==
int device_cu_count(int device) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 256;
    return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
}

static uint32_t device_cu_count_cached() {   // of the current device; asked once per device
    static int cached[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) return (uint32_t)device_cu_count(dev);
    if (cached[dev] == 0) cached[dev] = device_cu_count(dev);
    return (uint32_t)cached[dev];
}

static uint32_t persistent_blocks(uint64_t items, uint32_t block, uint32_t blocks_per_cu) {
    uint32_t blocks = device_cu_count_cached() * blocks_per_cu;
    const uint64_t needed = (items + block - 1) / block;
    if (needed < blocks) blocks = (uint32_t)needed;
    return blocks;
}

uint32_t stream_kernel_max_threads(uint32_t blocks_per_cu) {
    return device_cu_count_cached() * (blocks_per_cu ? blocks_per_cu : 8u) * 256u;  // k_trace / k_queue blocks are <= 256 threads
}

// Cam for a launch: shader.wgsl:690,702-708 in the shader's operation order.  Host code of this file is compiled
// with the same -ffp-contract=off, and +, -, *, /, sqrt of binary32 are correctly rounded on both sides, so these
// are the values the kernels used to work out for themselves (the parity suite compares every frame with the oracle).
static Cam host_cam(const rb_uniforms& u) {
    struct V { float x, y, z; };
    auto cross = [](V a, V b) { return V{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; };
    auto normalize = [](V a) {
        const float len = __builtin_sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z);
        return V{a.x / len, a.y / len, a.z / len};
    };
    // ranges of rb_device_math.hpp's rcp_safe / div_safe_den
    auto fast_den = [](float b) {
        uint32_t x;
        __builtin_memcpy(&x, &b, 4);
        x &= 0x7FFFFFFFu;
        return (x - 0x0D800000u) < 0x64000000u && (x - 0x21800000u) < 0x3C000000u && (x & 0x007FFFFFu) != 0x007FFFFFu;
    };
    Cam c{};
    c.aspect = (float)u.width / (float)u.height;
    const V fwd = normalize(V{u.camera.dir[0], u.camera.dir[1], u.camera.dir[2]});
    const V right = normalize(cross(V{0.0f, 1.0f, 0.0f}, fwd));
    const V up = cross(fwd, right);
    for (int i = 0; i < 3; i++) c.pos[i] = u.camera.pos[i];
    c.fwd[0] = fwd.x, c.fwd[1] = fwd.y, c.fwd[2] = fwd.z;
    c.right[0] = right.x, c.right[1] = right.y, c.right[2] = right.z;
    c.up[0] = up.x, c.up[1] = up.y, c.up[2] = up.z;
    c.fov = u.camera.pane_width / (2.0f * u.camera.pane_distance * c.aspect);
    c.wm1 = (float)(u.width - 1u);
    c.hm1 = (float)(u.height - 1u);
    c.fast_wh = (RB_FAST_DIV && fast_den(c.wm1) && fast_den(c.hm1)) ? 1u : 0u;
    c.inv_wm1 = c.fast_wh ? 1.0f / c.wm1 : 0.0f;
    c.inv_hm1 = c.fast_wh ? 1.0f / c.hm1 : 0.0f;
    return c;
}

int launch_render(const KParams& p_, uint32_t kernel, bool stats, void* stream_, LaunchInfo* info, void* ev_after_trace) {
    KParams p = p_;
    p.cam = host_cam(p.u);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    LaunchInfo li{};
    const size_t lds = (size_t)kStackEntryBytes * p.stack_depth * 256u;
    li.lds_bytes = lds;
    if (kernel == RB_KERNEL_PIXEL) {
        const uint32_t tiles_x = (p.u.width + 15u) / 16u, tiles_y = (p.local_rows + 15u) / 16u;
        li.grid = tiles_x * tiles_y;
        li.block = kPixelBlock;
        li.kernel_name = "k_pixel";
        if (li.grid == 0) return 0;
        if (stats)
            hipLaunchKernelGGL(k_pixel<true>, dim3(li.grid), dim3(li.block), lds, stream, p);
        else
            hipLaunchKernelGGL(k_pixel<false>, dim3(li.grid), dim3(li.block), lds, stream, p);
    } else if (kernel == RB_KERNEL_QUEUE) {
        const uint64_t items = (uint64_t)((p.u.width + 7u) / 8u) * ((p.local_rows + 7u) / 8u) * 64u;
        li.grid = persistent_blocks(items, kQueueBlock, p.blocks_per_cu ? p.blocks_per_cu : 8u);  // residency is set by VGPRs
        li.block = kQueueBlock;
        li.kernel_name = "k_queue";
        if (li.grid == 0) return 0;
        hipError_t e = hipMemsetAsync(p.queue, 0, sizeof(uint32_t), stream);
        if (e != hipSuccess) return (int)e;
        if (stats)
            hipLaunchKernelGGL(k_queue<true>, dim3(li.grid), dim3(li.block), lds, stream, p);
        else
            hipLaunchKernelGGL(k_queue<false>, dim3(li.grid), dim3(li.block), lds, stream, p);
    } else {
        const uint64_t tiles = (uint64_t)((p.u.width + 7u) / 8u) * ((p.local_rows + 7u) / 8u);
        const uint64_t items = tiles * 64u * p.n_passes * p.samples_per_pass;
        // which trace kernel, by what the runtime has prepared: the chunked walk (multi-node trees, the default), the
        // library's own tree (RB_FLAG_FAST_BVH), the reference-order walk (RB_FLAG_REFERENCE_WALK; from LDS when the
        // mesh fits next to the stacks: one 1024-thread block per CU, measured faster than three 256-thread blocks
        // with a copy each), the sphere tree, or the plain kernel (single-node tree, <= 64 spheres)
        enum Variant { PLAIN, BVH, BVH_LDS, FAST, SPH, CHUNK };
        const bool multi = p.u.bvh_node_count > 1u && !p.no_leaf_stepping;
        const size_t scene_lds = (size_t)p.u.bvh_node_count * 48u + (size_t)p.index_len * 48u;
        Variant v = PLAIN;
        if (multi && p.chunk_nodes != nullptr) v = CHUNK;
        else if (multi && p.fast_nodes != nullptr) v = FAST;
        else if (multi) {
            int dev = 0;
            (void)hipGetDevice(&dev);
            v = (p.lds_mode != 1u && lds * 4u + scene_lds <= max_dynamic_lds(dev)) ? BVH_LDS : BVH;
        }
        else if (p.sph_nodes != nullptr && !p.no_leaf_stepping) v = SPH;
        static const char* const names[] = {"k_trace", "k_trace_bvh", "k_trace_bvh_lds", "k_trace_fast", "k_trace_sph", "k_trace_chunk"};
        li.kernel_name = names[v];
        li.block = v == BVH_LDS ? 1024u : kTraceBlock;
        li.lds_bytes = v == BVH_LDS ? lds * 4u + scene_lds : v == CHUNK ? lds + (kTraceBlock / 64u) * kChunkWaveLds
                                     : v == SPH ? lds + (kTraceBlock / 64u) * kSphWaveLds : lds;
        // residency (registers): k_trace 6 waves/SIMD, the stepped walks 4
        // (a launch of one or two samples per pixel -- the progressive iterator's -- leaves a wavefront of the full grid a
        // few hundred items: with half the grid each regenerates paths for longer and the tail is shorter: 1080p, 1 spp,
        // 0.72 instead of 0.80 ms per frame)
        const uint32_t dense = (items >= (uint64_t)device_cu_count_cached() * 8u * 4u * 1024u) ? 8u : 4u;
        const uint32_t blocks_per_cu = v == BVH_LDS ? 1u : p.blocks_per_cu ? p.blocks_per_cu : v == FAST ? 4u : v == SPH ? (uint32_t)RB_SPH_WAVES
                                                                                                   : v == CHUNK ? (uint32_t)RB_CHUNK_WAVES : dense;
        li.grid = persistent_blocks(items, li.block, blocks_per_cu);
        if (li.grid == 0) return 0;
        // batch: the queue word sustains about 90 M atomics/s chip-wide, which 64-item reservations reach at 5-6 G
        // items/s (C1 and C2 run there): about 8 reservations per wave, 64..512 items, a multiple of 64
        // (profiles/r02_queue_batch.txt: C1 at 64 spp 17.1 -> 21.8 G segments/s)
        KParams q = p;
        const uint64_t waves = (uint64_t)li.grid * (li.block / 64u);
        // (the walks of trees and sphere sets hand out far fewer items per second: 64 reservations per wave, for balance)
        uint64_t batch = items / (waves * (v == PLAIN ? 8u : 64u));
        batch = (batch / 64u) * 64u;
        if (batch < 64u) batch = 64u;
        if (batch > (v == PLAIN ? 512u : 4096u)) batch = (v == PLAIN ? 512u : 4096u);
        // k_trace combines its colour stores per 64-item row through a ring of kRingRows rows (ColorRing), which needs
        // reservations of whole multiples of kRingRows rows; smaller launches keep their finer reservations and store directly
        if (v == PLAIN && batch >= 256u) batch = (batch / 256u) * 256u;
        if (p.queue_batch) batch = p.queue_batch;
        q.queue_batch = (uint32_t)batch;
        {   // reciprocals for the item -> (tile, sample) -> (tx, ty) divisions (udiv_magic)
            const uint64_t S = (uint64_t)p.n_passes * p.samples_per_pass, tiles_x = (p.u.width + 7u) / 8u;
            q.magic_S = S > 1u ? (uint32_t)((1ull << 32) / S) : 0xFFFFFFFFu;
            q.magic_tiles_x = tiles_x > 1u ? (uint32_t)((1ull << 32) / tiles_x) : 0xFFFFFFFFu;
        }
        // the queue starts behind the waves' own first reservations (ItemQueue); the walks of trees and sphere sets get one band of
        // items and one queue word per group of blocks that share an XCD
        hipError_t e = hipSuccess;
        // (stripes of about four tile rows, whole multiples of the batch; at least four stripes per band, or one word as before)
        const uint64_t row_items = (uint64_t)((p.u.width + 7u) / 8u) * 64u * p.n_passes * p.samples_per_pass;
        const uint64_t stripe = std::max<uint64_t>(((4u * row_items + batch - 1u) / batch) * batch, 8u * batch);
        const bool banded = v != PLAIN && RB_XCD_BANDS && li.grid >= kQueueGroups && items >= 4u * kQueueGroups * stripe && stripe < (1ull << 30);
        if (banded) {
            q.queue_groups = kQueueGroups;
            q.queue_region = (uint32_t)stripe;
            QueueInit qi;
            for (uint32_t g = 0; g < kQueueGroups; g++) {
                const uint64_t blocks_g = (li.grid + kQueueGroups - 1u - g) / kQueueGroups;
                qi.start[g] = (uint32_t)std::min<uint64_t>(blocks_g * (li.block / 64u) * batch, 0x7FFFFFFFull);   // band-local
            }
            hipLaunchKernelGGL(k_queue_init, dim3(1), dim3(64), 0, stream, p.queue, qi);
            e = hipGetLastError();
        } else {
            q.queue_groups = 1u;
            q.queue_region = 0u;
            const uint64_t queue_start = waves * batch;   // <= 8192 waves * 4096 items
            e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(p.queue), (int)(uint32_t)queue_start, 1, stream);
        }
        if (e != hipSuccess) return (int)e;
        const dim3 grid(li.grid), block(li.block);
        switch (v) {
            case SPH:
                if (stats) hipLaunchKernelGGL(k_trace_sph<true>, grid, block, li.lds_bytes, stream, q);
                else hipLaunchKernelGGL(k_trace_sph<false>, grid, block, li.lds_bytes, stream, q);
                break;
            case CHUNK:
                if (p.sph_nodes != nullptr) {
                    if (stats) hipLaunchKernelGGL((k_trace_chunk<true, true>), grid, block, li.lds_bytes, stream, q);
                    else hipLaunchKernelGGL((k_trace_chunk<false, true>), grid, block, li.lds_bytes, stream, q);
                } else {
                    if (stats) hipLaunchKernelGGL((k_trace_chunk<true, false>), grid, block, li.lds_bytes, stream, q);
                    else hipLaunchKernelGGL((k_trace_chunk<false, false>), grid, block, li.lds_bytes, stream, q);
                }
                break;
            case FAST:
                if (stats) hipLaunchKernelGGL(k_trace_fast<true>, grid, block, lds, stream, q);
                else hipLaunchKernelGGL(k_trace_fast<false>, grid, block, lds, stream, q);
                break;
            case BVH_LDS:
                e = stats ? hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trace_bvh<true, true, 1024u>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)li.lds_bytes)
                          : hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trace_bvh<false, true, 1024u>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)li.lds_bytes);
                if (e != hipSuccess) return (int)e;
                if (stats) hipLaunchKernelGGL((k_trace_bvh<true, true, 1024u>), grid, block, li.lds_bytes, stream, q);
                else hipLaunchKernelGGL((k_trace_bvh<false, true, 1024u>), grid, block, li.lds_bytes, stream, q);
                break;
            case BVH:
                if (stats) hipLaunchKernelGGL((k_trace_bvh<true, false, kTraceBlock>), grid, block, lds, stream, q);
                else hipLaunchKernelGGL((k_trace_bvh<false, false, kTraceBlock>), grid, block, lds, stream, q);
                break;
            case PLAIN:
                if (p.u.bvh_node_count > 1u || p.sph_nodes != nullptr) {   // per-segment ablation of a multi-node tree or of the sphere tree
                    if (stats) hipLaunchKernelGGL((k_trace<true, true>), grid, block, lds, stream, q);
                    else hipLaunchKernelGGL((k_trace<false, true>), grid, block, lds, stream, q);
                } else {
                    if (stats) hipLaunchKernelGGL((k_trace<true, false>), grid, block, lds, stream, q);
                    else if (items >= kTraceManyItems) hipLaunchKernelGGL((k_trace<false, false, RB_TRACE_WAVES_BIG>), grid, block, lds, stream, q);
                    else hipLaunchKernelGGL((k_trace<false, false>), grid, block, lds, stream, q);
                }
                break;
        }
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
        if (ev_after_trace) (void)hipEventRecord(static_cast<hipEvent_t>(ev_after_trace), stream);
        hipLaunchKernelGGL(k_accumulate, dim3((uint32_t)((tiles + 3u) / 4u)), dim3(256), 0, stream, p);
    }
    if (info) *info = li;
    return (int)hipGetLastError();
}

int launch_prep_tris(const rb_gpu_triangle* tris, uint32_t tri_count, const uint32_t* indices,
                     uint32_t index_len, PrepTri* out, PrepTriShade* shade, void* stream_) {
    if (index_len == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const uint32_t block = 256, grid = (index_len + block - 1) / block;
    hipLaunchKernelGGL(k_prep_tris, dim3(grid), dim3(block), 0, stream, tris, tri_count, indices, index_len, out,
                       shade);
    return (int)hipGetLastError();
}

int launch_prep_materials(void* first_material, uint32_t stride, uint32_t n, void* stream_) {
    if (n == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(k_prep_materials, dim3((n + 255) / 256), dim3(256), 0, stream,
                       static_cast<unsigned char*>(first_material), stride, n);
    return (int)hipGetLastError();
}

int launch_gather_tris(const PrepTri* ptris, const uint32_t* slots, uint32_t n, PrepTri* out, void* stream_) {
    if (n == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(k_gather_tris, dim3((n + 255) / 256), dim3(256), 0, stream, ptris, slots, n, out);
    return (int)hipGetLastError();
}

int launch_chunk_gather(const PrepTri* ptris, const uint32_t* pos_slot, const uint32_t* pos_rank, uint32_t n, float* a,
                        float* b, float* c, void* stream_) {
    if (n == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(k_chunk_gather, dim3((n + 255) / 256), dim3(256), 0, stream, ptris, pos_slot, pos_rank, n,
                       reinterpret_cast<float4*>(a), reinterpret_cast<float4*>(b), reinterpret_cast<float4*>(c));
    return (int)hipGetLastError();
}

int launch_deinterleave(const uint32_t* gathered, uint32_t* frame, uint32_t width, uint32_t height, uint32_t padded_rows,
                        uint32_t stripe_rows, uint32_t shard_count, void* stream_) {
    if (width == 0 || height == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(k_deinterleave, dim3((width + 255u) / 256u, height < 65535u ? height : 65535u), dim3(256), 0, stream, gathered,
                       frame, width, height, padded_rows, stripe_rows, shard_count);
    return (int)hipGetLastError();
}

size_t max_dynamic_lds(int device) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess || v <= 0) return 64u * 1024u;
    return (size_t)v;
}

int launch_div_exhaustive(uint32_t b_begin, uint32_t b_count, uint32_t ea, uint32_t eb, uint32_t a_begin,
                          uint32_t a_count, unsigned long long* mismatch, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(k_div_exhaustive, dim3((b_count + 255) / 256), dim3(256), 0, stream, b_begin, ea, eb, a_begin,
                       a_count, mismatch);
    return (int)hipGetLastError();
}

int launch_rcp_exhaustive(uint32_t expo, uint32_t* mismatch, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(k_rcp_exhaustive, dim3((1u << 23) / 256), dim3(256), 0, stream, expo, mismatch);
    return (int)hipGetLastError();
}

// -> lane accesses per second (0 on failure): the best of a few shapes (8 or 16 loads in flight per lane, 4 or 8 waves
// per SIMD); table_bytes is rounded down to a power of two >= 8 KiB
double measure_l1_gather(size_t table_bytes, uint32_t rounds) {
    size_t lines = 64;
    while (lines * 2 * 128 <= table_bytes) lines *= 2;
    float4 *table = nullptr, *sink = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&table), lines * 128) != hipSuccess) return 0.0;
    if (hipMalloc(reinterpret_cast<void**>(&sink), 16) != hipSuccess) { (void)hipFree(table); return 0.0; }
    (void)hipMemset(table, 0, lines * 128);
    hipEvent_t t0, t1;
    (void)hipEventCreate(&t0);
    (void)hipEventCreate(&t1);
    double best = 0.0;
    for (int shape = 0; shape < 4; shape++) {
        const uint32_t inflight = (shape & 1) ? 16u : 8u, blocks = device_cu_count_cached() * ((shape & 2) ? 8u : 4u);
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(t0, nullptr);
            if (inflight == 16u) hipLaunchKernelGGL(k_l1_gather<16>, dim3(blocks), dim3(256), 0, nullptr, table, (uint32_t)(lines - 1), rounds, sink);
            else hipLaunchKernelGGL(k_l1_gather<8>, dim3(blocks), dim3(256), 0, nullptr, table, (uint32_t)(lines - 1), rounds, sink);
            (void)hipEventRecord(t1, nullptr);
            if (hipEventSynchronize(t1) != hipSuccess) break;
            float ms = 0.0f;
            (void)hipEventElapsedTime(&ms, t0, t1);
            if (ms > 0.0f) best = std::max(best, (double)blocks * 256.0 * rounds * inflight / (ms * 1e-3));
        }
    }
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    (void)hipFree(table);
    (void)hipFree(sink);
    return best;
}

// pass-occupancy counters of a profiling build (tools/ablate/rb_profile.patch defines RB_WALK_PROFILE and the counting);
// the product build counts nothing
#ifndef RB_WALK_PROFILE
int debug_walk_profile(unsigned long long*, int) { return -1; }
#endif

int launch_debug_math(const float* a, const float* b, float* out, uint32_t n, void* stream_) {
    if (n == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(k_debug_math, dim3((n + 255) / 256), dim3(256), 0, stream, a, b, out, n);
    return (int)hipGetLastError();
}

}  // namespace rb
