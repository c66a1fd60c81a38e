// Internal (non-ABI) interface between assembly_loss.hip and grid_nn.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpa {

// extra workspace the grid-pruned whole-shape search needs (floats / int32s)
int64_t grid_workspace_floats(int64_t B, int64_t P, int64_t N);
int64_t grid_workspace_ints(int64_t B);
// inside that scratch: the per-part bounding boxes [B*P][12] (lo of shape 1, lo of shape 2, hi of shape 1, hi of shape 2)
// and the one-word ticket of the sort launch — both written by the producer of S1 / S2 BEFORE launch_grid_shape_search
// (boxes of the valid parts; ticket = 0).
float* grid_bbox(float* fws, int64_t B, int64_t P, int64_t N);
unsigned* grid_ticket(int32_t* iws, int64_t B);

// Exact NN of every valid point of S1 in S2 and vice versa (whole shapes of each sample, padded parts as
// one representative).  Writes idx1/idx2 [B,P,N] (index within the sample, -1 if none) and the per-part
// distance sums into tile_sums[dir][m*tiles + 0].  fws/iws: scratch sized by the functions above.
// before_search / after_search (nullable) are recorded around the search kernel proper.
// route (nullable) [B]: samples with route[b] == 0 are left to the leaf search (no work, no sums written for them).
// phases: 1 = build the grid, 2 = the search kernel, 4 = per-part distance sums (a caller that puts another search's
// launches between the search kernel and the sums calls three times).
int launch_grid_shape_search(const float* valids, const float* S1, const float* S2, int64_t B, int64_t P,
                             int64_t N, int tiles, float* fws, int32_t* iws, int32_t* idx1, int32_t* idx2,
                             float* tile_sums, hipEvent_t before_search, hipEvent_t after_search, hipStream_t s,
                             const int* route = nullptr, int phases = 7);

// ---- the same exact pruned search for two plain clouds per sample (the generic operator, chamfer.hip) -----------------
// xyz1 [B, n1, 3], xyz2 [B, n2, 3] -> dist / idx of mpa_chamfer_forward's contract, bit for bit, for every sample whose
// coordinates are finite and <= 1e15 in magnitude; the other samples are left untouched and flagged in (*fallback)[b]
// (device memory inside `workspace`) for the caller's exhaustive scan.  `workspace`: cloud_grid_workspace_bytes() bytes.
int64_t cloud_grid_workspace_bytes(int64_t B, int64_t n1, int64_t n2);
bool cloud_grid_supported(int64_t B, int64_t n1, int64_t n2);
int launch_cloud_grid_search(const float* xyz1, const float* xyz2, int64_t B, int64_t n1, int64_t n2, float* dist1,
                             int64_t* idx1, float* dist2, int64_t* idx2, void* workspace, const int** fallback,
                             hipStream_t s);
// ... and, behind the caller's scan of the flagged samples: points that repeat their predecessor were not searched (neither
// here nor — same answer — need they be): they take the answer of the first point of their run.
void launch_cloud_copy_runs(int64_t B, int64_t n1, int64_t n2, float* dist1, int64_t* idx1, float* dist2, int64_t* idx2,
                            void* workspace, hipStream_t s);

// ---- leaf search (leaf_nn.hip): the Chamfer searches of the fused loss over per-part k-d leaves -------------------------------
// One transformed cloud as the pose kernel leaves it: records [B*P][Npad] float4 in the parts' k-d order, one box per
// leaf [B*P][Npad / 32][8] and per part [B*P][8] (lo xyz, -, hi xyz, -), and the cloud in original order [B, P, N, 3].
struct LeafCloud {
  const float* rec;
  const float* leaf;
  const float* part;
  const float* orig;
};
int leaf_npad(int64_t N);                  // slots per part: the power of two >= max(N, 32)
bool leaf_supported(int64_t P, int64_t N);  // P <= 64 parts, N <= 2048 points per part
// k-d order of every valid part's points: sorted [B*P][Npad] float4 (local x, y, z, original index | -1)
void launch_leaf_order(const float* part_pcs, const float* valids, int64_t B, int64_t P, int64_t N, float* sorted,
                       hipStream_t s);
// exact NN of cloud A's valid points in cloud B (idx1) and vice versa (idx2): shape = every point against the sample's whole
// other shape (indices p * N + n), else every part against its own copy (indices n); per-wave distance sums into
// wave_sums[dir][m][NW], NW = max(1, Npad / 64).  scratch: leaf_scratch_floats() floats (16-byte aligned), shared by the two
// searches of a loss evaluation; its two counters (leaf_heavy_counters) must be zero when the first search starts.
// route (nullable) [B]: the whole-shape search skips samples with route[b] != 0 (searched by the grid, grid_nn.hip).
int64_t leaf_scratch_floats(int64_t B, int64_t P, int64_t N);
int* leaf_heavy_counters(float* scratch);
int* leaf_route(float* scratch);  // [B] ints inside the scratch
void launch_leaf_search(bool shape, const float* valids, const LeafCloud& A, const LeafCloud& B_, int64_t B, int64_t P,
                        int64_t N, int32_t* idx1, int32_t* idx2, float* wave_sums, float* scratch, hipStream_t s,
                        const int* route = nullptr);
// Which search answers each sample's whole-shape term (route[b]: 1 = grid, 0 = leaf).  force < 0: by the sample itself —
// the share of the ground-truth shape's bounding box that its parts' boxes fill (part boxes pbox [B*P][8]); many small
// parts in a large box (a clumpy cloud) is where a uniform grid loses and the leaves win.  force 0 / 1: every sample alike.
void launch_leaf_route(const float* valids, const float* pbox, int64_t B, int64_t P, int force, int* route, hipStream_t s);

// ---- matrix-core gated exact NN between two mid-sized clouds (gate_nn.hip) ---------------------------------------------------
// Per-part Chamfer of the fused loss: part m of C1 against part m of C2 (both [B, P, N, 3]; padded parts skipped), arg-mins
// into idx1 / idx2 [B, P, N] and per-block distance sums into tile_sums[dir][m][gate_tiles(N, N)].
int gate_tiles(int64_t na, int64_t nb);
bool gate_supported(int64_t na, int64_t nb);
void launch_gate_part_search(const float* valids, const float* C1, const float* C2, int64_t B, int64_t P, int64_t N,
                             int32_t* idx1, int32_t* idx2, float* tile_sums, hipStream_t s);
// The generic operator's contract (mpa_chamfer_forward) for xyz1 [batch, n1, 3], xyz2 [batch, n2, 3]; no workspace.
void launch_gate_cloud_search(const float* xyz1, const float* xyz2, int64_t batch, int64_t n1, int64_t n2, float* dist1,
                              int64_t* idx1, float* dist2, int64_t* idx2, hipStream_t s);

}  // namespace mpa
