#!/bin/bash
# Host-only timing of the chunked walk's tree build (rb_bvh.cpp chunk_tree_build) on the C5 mesh, on the machine it runs on:
# parallel (one thread per granted CPU) and RB_HOST_BUILD_SEQUENTIAL=1.  No GPU involved.
set -e
cd "$(dirname "$0")/.."
python3 - <<'PY'
import numpy as np
from renderbaby_amd import scenes
s = scenes.mesh_c5().with_params(width=64, height=64, spp=1)
np.ascontiguousarray(s.bvh_triangles).tofile('/tmp/c5_tris.bin'); np.ascontiguousarray(s.bvh_nodes).tofile('/tmp/c5_nodes.bin'); np.ascontiguousarray(s.bvh_indices).tofile('/tmp/c5_idx.bin')
PY
cat > /tmp/tb.cpp <<'CPP'
#include "renderbaby_amd/csrc/rb_internal.hpp"
#include <chrono>
#include <cstdio>
#include <fstream>
template <class T> std::vector<T> rd(const char* p) { std::ifstream f(p, std::ios::binary | std::ios::ate); size_t n = f.tellg(); f.seekg(0); std::vector<T> v(n / sizeof(T)); f.read((char*)v.data(), n); return v; }
int main() {
    auto tris = rd<rb_gpu_triangle>("/tmp/c5_tris.bin"); auto nodes = rd<rb_bvh_node>("/tmp/c5_nodes.bin"); auto idx = rd<uint32_t>("/tmp/c5_idx.bin");
    for (int r = 0; r < 4; r++) {
        rb::ChunkTree t;
        auto t0 = std::chrono::steady_clock::now();
        bool ok = rb::chunk_tree_build(tris.data(), tris.size(), idx.data(), idx.size(), nodes.data(), nodes.size(), 32, t);
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("ok %d nodes %zu positions %zu depth %u: %.1f ms\n", ok, t.nodes.size(), t.pos_slot.size(), t.depth, ms);
    }
}
CPP
g++ -O2 -std=c++17 -pthread -I. -o /tmp/tb /tmp/tb.cpp renderbaby_amd/csrc/rb_bvh.cpp
echo "parallel ($(nproc) CPUs visible):"; /tmp/tb
echo "sequential:"; RB_HOST_BUILD_SEQUENTIAL=1 /tmp/tb
