// Host-side feeder: bi-directional edge list -> per-half CSR by destination (+ folded norms).
// Stands where data_loader.py:132-157 + model.py:72-80,88-97 stand in the reference. Integer work,
// O(E), single pass per half; the result is the order a CPU scatter-add visits the edges, so device
// sums are reproducible and comparable term by term with the oracle.
#include <cmath>
#include <cstring>
#include <vector>

#include "mgcn_common.h"

namespace mgcn {
char *error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}
}  // namespace mgcn

extern "C" int mgcn_abi_version(void) { return MGCN_ABI_VERSION; }
extern "C" const char *mgcn_last_error(void) { return mgcn::error_buffer(); }

extern "C" int mgcn_csr_build_host(int64_t num_nodes, int64_t num_edges_half, int64_t num_rel_rows,
                                   const int64_t *edge_index_host, const int64_t *edge_type_host,
                                   int32_t *rowptr_host, mgcn_edge_rec *rec_host, int64_t *perm_host,
                                   int32_t *slot_dst_host, int32_t *srcptr_host, int32_t *srcslots_host,
                                   int32_t *typeptr_host, int32_t *typeslots_host) {
  const int64_t N = num_nodes, E = num_edges_half, E2 = 2 * num_edges_half;
  MGCN_REQUIRE(N >= 0 && E >= 0 && num_rel_rows >= 0, "csr_build: negative size");
  MGCN_REQUIRE(N < (int64_t(1) << 31) - 1 && E2 < (int64_t(1) << 31) - 1, "csr_build: sizes exceed int32 slots");
  MGCN_REQUIRE(rowptr_host && (E == 0 || (rec_host && perm_host)), "csr_build: null output");
  MGCN_REQUIRE(E == 0 || (edge_index_host && edge_type_host), "csr_build: null input");
  const int64_t *src = edge_index_host, *dst = edge_index_host + E2;
  for (int64_t e = 0; e < E2; ++e) {
    MGCN_REQUIRE(src[e] >= 0 && src[e] < N && dst[e] >= 0 && dst[e] < N,
                 "csr_build: edge %lld endpoint (%lld -> %lld) outside [0, %lld)", (long long)e,
                 (long long)src[e], (long long)dst[e], (long long)N);
    MGCN_REQUIRE(edge_type_host[e] >= 0 && edge_type_host[e] < num_rel_rows,
                 "csr_build: edge %lld type %lld outside [0, %lld)", (long long)e,
                 (long long)edge_type_host[e], (long long)num_rel_rows);
  }
  std::vector<float> cinv(N);
  std::vector<int32_t> cursor(N + 1);
  const bool bwd = srcptr_host != nullptr;
  MGCN_REQUIRE(bwd == (typeptr_host != nullptr) &&
                   (E == 0 || (bwd == (slot_dst_host != nullptr) && bwd == (srcslots_host != nullptr) &&
                               bwd == (typeslots_host != nullptr))),
               "csr_build: backward index outputs must be given all together or not at all");
  for (int h = 0; h < 2; ++h) {
    const int64_t lo = h * E;
    int32_t *rowptr = rowptr_host + h * (N + 1);
    // degree by SOURCE within the half (model.py:74-75), deg^-1/2 with inf -> 0 (model.py:76-77)
    std::vector<int32_t> deg(N, 0);
    for (int64_t e = lo; e < lo + E; ++e) deg[src[e]]++;
    for (int64_t n = 0; n < N; ++n) cinv[n] = deg[n] ? 1.0f / std::sqrt(static_cast<float>(deg[n])) : 0.0f;
    // counting sort by destination, stable in edge id
    std::memset(rowptr, 0, sizeof(int32_t) * (N + 1));
    for (int64_t e = lo; e < lo + E; ++e) rowptr[dst[e] + 1]++;
    for (int64_t n = 0; n < N; ++n) rowptr[n + 1] += rowptr[n];
    std::memcpy(cursor.data(), rowptr, sizeof(int32_t) * (N + 1));
    for (int64_t e = lo; e < lo + E; ++e) {
      const int64_t slot = lo + cursor[dst[e]]++;
      mgcn_edge_rec r;
      r.src = static_cast<int32_t>(src[e]);
      r.type = static_cast<int32_t>(edge_type_host[e]);
      r.norm = cinv[src[e]] * cinv[dst[e]];  // model.py:78 (edge_weight == 1)
      r.eid = static_cast<int32_t>(e);
      rec_host[slot] = r;
      perm_host[slot] = e;
      if (bwd) slot_dst_host[slot] = static_cast<int32_t>(dst[e]);
    }
    if (bwd) {  // slots of this half grouped by source, ascending slot id
      int32_t *srcptr = srcptr_host + h * (N + 1);
      std::memset(srcptr, 0, sizeof(int32_t) * (N + 1));
      for (int64_t s = lo; s < lo + E; ++s) srcptr[rec_host[s].src + 1]++;
      for (int64_t n = 0; n < N; ++n) srcptr[n + 1] += srcptr[n];
      std::memcpy(cursor.data(), srcptr, sizeof(int32_t) * (N + 1));
      for (int64_t s = lo; s < lo + E; ++s) srcslots_host[lo + cursor[rec_host[s].src]++] = static_cast<int32_t>(s);
    }
  }
  if (bwd) {  // all slots grouped by relation row, ascending slot id
    std::vector<int32_t> tcur(num_rel_rows + 1, 0);
    std::memset(typeptr_host, 0, sizeof(int32_t) * (num_rel_rows + 1));
    for (int64_t s = 0; s < E2; ++s) typeptr_host[rec_host[s].type + 1]++;
    for (int64_t t = 0; t < num_rel_rows; ++t) typeptr_host[t + 1] += typeptr_host[t];
    std::memcpy(tcur.data(), typeptr_host, sizeof(int32_t) * (num_rel_rows + 1));
    for (int64_t s = 0; s < E2; ++s) typeslots_host[tcur[rec_host[s].type]++] = static_cast<int32_t>(s);
  }
  return MGCN_OK;
}
