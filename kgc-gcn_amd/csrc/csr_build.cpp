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
                                   int64_t hub_threshold, int64_t hub_chunk, int32_t *rowptr_host,
                                   mgcn_edge_rec *rec_host, int64_t *perm_host, int32_t *hubinfo_host,
                                   int32_t *chunks_host, int64_t max_chunks, int64_t *num_chunks_host,
                                   int32_t *slot_dst_host, int32_t *mirror_host, int32_t *typeptr_host,
                                   int32_t *typeslots_host) {
  const int64_t N = num_nodes, E = num_edges_half, E2 = 2 * num_edges_half;
  MGCN_REQUIRE(N >= 0 && E >= 0 && num_rel_rows >= 0, "csr_build: negative size");
  MGCN_REQUIRE(N < (int64_t(1) << 31) - 1 && E2 < (int64_t(1) << 31) - 1, "csr_build: sizes exceed int32 slots");
  MGCN_REQUIRE(rowptr_host && (E == 0 || (rec_host && perm_host)), "csr_build: null output");
  MGCN_REQUIRE(E == 0 || (edge_index_host && edge_type_host), "csr_build: null input");
  const bool hubs = hub_threshold > 0;
  MGCN_REQUIRE(!hubs || (hub_chunk > 0 && hubinfo_host && chunks_host && num_chunks_host),
               "csr_build: hub splitting needs hub_chunk > 0 and the hubinfo / chunks / num_chunks outputs");
  const bool bwd = typeptr_host != nullptr;
  MGCN_REQUIRE(E == 0 || (bwd == (slot_dst_host != nullptr) && bwd == (mirror_host != nullptr) &&
                          bwd == (typeslots_host != nullptr)),
               "csr_build: backward index outputs must be given all together or not at all");
  const int64_t *src = edge_index_host, *dst = edge_index_host + E2;
  for (int64_t e = 0; e < E2; ++e) {
    MGCN_REQUIRE(src[e] >= 0 && src[e] < N && dst[e] >= 0 && dst[e] < N,
                 "csr_build: edge %lld endpoint (%lld -> %lld) outside [0, %lld)", (long long)e,
                 (long long)src[e], (long long)dst[e], (long long)N);
    // the LAST row of the relation table is the self-loop row (model.py:86,91): only the self-loop pass uses it
    MGCN_REQUIRE(edge_type_host[e] >= 0 && edge_type_host[e] < num_rel_rows - 1,
                 "csr_build: edge %lld type %lld outside [0, %lld) (row %lld is the self-loop row)", (long long)e,
                 (long long)edge_type_host[e], (long long)(num_rel_rows - 1), (long long)(num_rel_rows - 1));
  }
  // ---- pass 1: destination counts per half; a destination with more than hub_threshold slots in a half is a hub
  std::vector<int32_t> cnt(2 * N, 0);
  for (int h = 0; h < 2; ++h)
    for (int64_t e = h * E; e < (h + 1) * E; ++e) cnt[h * N + dst[e]]++;
  auto is_hub = [&](int h, int64_t n) { return hubs && cnt[h * N + n] > hub_threshold; };
  // ---- slot layout: [in-half non-hub segments | out-half non-hub segments | hub segments (node, half order)].
  // rowptr holds ABSOLUTE slot positions of the non-hub segments (a hub's segment there is empty).
  int64_t pos = 0;
  for (int h = 0; h < 2; ++h) {
    int32_t *rowptr = rowptr_host + h * (N + 1);
    for (int64_t n = 0; n < N; ++n) {
      rowptr[n] = int32_t(pos);
      if (!is_hub(h, n)) pos += cnt[h * N + n];
    }
    rowptr[N] = int32_t(pos);
  }
  std::vector<int32_t> cursor(2 * N);
  for (int h = 0; h < 2; ++h)
    for (int64_t n = 0; n < N; ++n) cursor[h * N + n] = rowptr_host[h * (N + 1) + n];
  int64_t nchunks = 0;
  if (hubinfo_host)
    for (int64_t i = 0; i < 4 * N; ++i) hubinfo_host[i] = (i & 1) ? 0 : -1;  // (first chunk, chunk count) = (-1, 0)
  if (hubs) {
    for (int64_t n = 0; n < N; ++n)       // node-major: the hubs of a destination range own ONE run of slots / chunks
      for (int h = 0; h < 2; ++h) {
        if (!is_hub(h, n)) continue;
        const int64_t len = cnt[h * N + n], nch = (len + hub_chunk - 1) / hub_chunk;
        MGCN_REQUIRE(nchunks + nch <= max_chunks, "csr_build: chunk table too small (%lld)", (long long)max_chunks);
        hubinfo_host[(h * N + n) * 2 + 0] = int32_t(nchunks);
        hubinfo_host[(h * N + n) * 2 + 1] = int32_t(nch);
        for (int64_t k = 0; k < nch; ++k) {
          int32_t *c = chunks_host + (nchunks + k) * 4;   // {begin, end, first chunk of the hub, chunks of the hub}
          c[0] = int32_t(pos + k * hub_chunk);
          c[1] = int32_t(pos + (k + 1 < nch ? (k + 1) * hub_chunk : len));
          c[2] = int32_t(nchunks);
          c[3] = int32_t(nch);
        }
        nchunks += nch;
        cursor[h * N + n] = int32_t(pos);
        pos += len;
      }
  }
  if (num_chunks_host) *num_chunks_host = nchunks;
  // ---- pass 2: stable fill in edge-id order; norms folded into the records
  std::vector<float> cinv(N);
  for (int h = 0; h < 2; ++h) {
    const int64_t lo = h * E;
    // degree by SOURCE within the half (model.py:74-75), deg^-1/2 with inf -> 0 (model.py:76-77)
    std::vector<int32_t> deg(N, 0);
    for (int64_t e = lo; e < lo + E; ++e) deg[src[e]]++;
    for (int64_t n = 0; n < N; ++n) cinv[n] = deg[n] ? 1.0f / std::sqrt(static_cast<float>(deg[n])) : 0.0f;
    for (int64_t e = lo; e < lo + E; ++e) {
      const int64_t slot = cursor[h * N + dst[e]]++;
      mgcn_edge_rec r;
      r.src = static_cast<int32_t>(src[e]);
      r.type = static_cast<int32_t>(edge_type_host[e]);
      r.norm = cinv[src[e]] * cinv[dst[e]];  // model.py:78 (edge_weight == 1)
      r.eid = static_cast<int32_t>(e);
      rec_host[slot] = r;
      perm_host[slot] = e;
      if (bwd) slot_dst_host[slot] = static_cast<int32_t>(dst[e]) | (h ? int32_t(0x80000000u) : 0);  // bit 31 = half
    }
  }
  if (bwd) {
    // mirror: slot of edge e <-> slot of its reverse edge (e + E) mod 2E. The edges that LEAVE node n in half h are
    // the reverses of the edges that ENTER n in half 1-h, so the by-source sums of the backward walk the same
    // destination runs (and hub chunks) as the forward, through this map.
    // That only holds for an edge list whose second half is the first half reversed (data_loader.py:143-149 builds
    // it so). The operator seam (model.py:82-101) accepts ANY [2, 2E] list split in halves by position: for one that
    // is not mirror-symmetric the forward arrays above are still exact, but there is no such map — mirror is filled
    // with -1 and the gradient w.r.t. x is refused by the caller (mgcn_aggregate_bwd documents it).
    bool mirrored = true;
    for (int64_t e = 0; e < E && mirrored; ++e) mirrored = src[e + E] == dst[e] && dst[e + E] == src[e];
    if (mirrored) {
      std::vector<int32_t> slot_of(E2);
      for (int64_t s = 0; s < E2; ++s) slot_of[perm_host[s]] = int32_t(s);
      for (int64_t s = 0; s < E2; ++s) mirror_host[s] = slot_of[(perm_host[s] + E) % E2];
    } else {
      for (int64_t s = 0; s < E2; ++s) mirror_host[s] = -1;
    }
    // all slots grouped by relation row, ascending slot id
    std::vector<int32_t> tcur(num_rel_rows + 1, 0);
    std::memset(typeptr_host, 0, sizeof(int32_t) * (num_rel_rows + 1));
    for (int64_t s = 0; s < E2; ++s) typeptr_host[rec_host[s].type + 1]++;
    for (int64_t t = 0; t < num_rel_rows; ++t) typeptr_host[t + 1] += typeptr_host[t];
    std::memcpy(tcur.data(), typeptr_host, sizeof(int32_t) * (num_rel_rows + 1));
    for (int64_t s = 0; s < E2; ++s) typeslots_host[tcur[rec_host[s].type]++] = static_cast<int32_t>(s);
  }
  return MGCN_OK;
}
