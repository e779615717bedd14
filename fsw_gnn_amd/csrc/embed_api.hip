// C-ABI entry points of the fused neighbourhood embed: argument checks and per-degree-class dispatch.
#include "fsw_common.h"

namespace fsw {
int launch_unit_table(const float* freqs, int S, int max_deg, float* table, int64_t ldt, hipStream_t stream);
int launch_zero_rows(const fsw_embed_args& a, hipStream_t stream);
int launch_embed_reg(const fsw_embed_args& a, bool unit_fast, int64_t rows_upper, hipStream_t stream);
int launch_embed_mid(const fsw_embed_args& a, bool unit_fast, int64_t rows_upper, hipStream_t stream);
int launch_embed_lds(const fsw_embed_args& a, int64_t rows_upper, hipStream_t stream);
int launch_embed_hub(const fsw_embed_args& a, int64_t rows_upper, hipStream_t stream);
int launch_embed_giant(const fsw_embed_args& a, int64_t rows_upper, hipStream_t stream);
int launch_embed_global(const fsw_embed_args& a, int64_t rows_upper, hipStream_t stream);
size_t embed_global_scratch_bytes(int64_t max_degree);
}  // namespace fsw

using namespace fsw;

extern "C" size_t fsw_unit_table_rows(int max_deg) { return (size_t)max_deg * (max_deg + 1) / 2; }

extern "C" int fsw_unit_coeff_table(const float* freqs, int S, int max_deg, float* table, int64_t ldt, fsw_stream_t stream) {
  FSW_REQUIRE(freqs && table, "fsw_unit_coeff_table: null pointer");
  FSW_REQUIRE(S >= 1 && ldt >= S && max_deg >= 1 && max_deg <= FSW_REG_MAX_DEG,
              "fsw_unit_coeff_table: need S >= 1, ldt >= S, 1 <= max_deg <= %d", FSW_REG_MAX_DEG);
  return launch_unit_table(freqs, S, max_deg, table, ldt, reinterpret_cast<hipStream_t>(stream));
}

extern "C" size_t fsw_embed_scratch_bytes(int64_t max_degree) {
  return max_degree > FSW_LDS_MAX_DEG ? embed_global_scratch_bytes(max_degree) : 0;
}

extern "C" int fsw_embed_f32(const fsw_embed_args* args, fsw_stream_t stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  FSW_REQUIRE(args, "fsw_embed_f32: null args");
  const fsw_embed_args& a = *args;
  FSW_REQUIRE(a.rowptr && a.perm && a.bin_start && a.Xp && a.freqs && a.out, "fsw_embed_f32: null pointer");
  FSW_REQUIRE(a.num_rows >= 1 && a.S >= 1 && a.ldp >= a.S && a.ldo >= a.S + a.has_mass, "fsw_embed_f32: bad sizes");
  FSW_REQUIRE(a.tau > 0.f, "fsw_embed_f32: total_mass_pad_thresh must be positive");  // reference fsw_embedding.py:637
  FSW_REQUIRE(a.has_mass == 0 || a.has_mass == 1, "fsw_embed_f32: has_mass must be 0 or 1");
  FSW_REQUIRE(a.mass_fn >= 0 && a.mass_fn <= 2, "fsw_embed_f32: mass_fn must be 0, 1 or 2");
  FSW_REQUIRE(!a.efeat || (a.w && a.Ve && a.d_edge >= 1 && a.ldve >= a.d_edge),
              "fsw_embed_f32: edge features need a coalesced weighted graph (fsw_graph_build_coalesced) and Ve");
  const bool unit_fast = (a.w == nullptr) && (a.tau <= 1.f);
  FSW_REQUIRE(!unit_fast || (a.unit_table && a.ldt >= a.S), "fsw_embed_f32: unit weights with tau <= 1 need unit_table");

  const int64_t nz = a.num_zero_rows < 0 ? a.num_rows : a.num_zero_rows;
  const int64_t nreg = a.num_reg_rows < 0 ? a.num_rows : a.num_reg_rows;
  const int64_t nlds = a.num_lds_rows < 0 ? a.num_rows : a.num_lds_rows;
  const int64_t nglob = a.num_global_rows < 0 ? a.num_rows : a.num_global_rows;
  int rc;
  if (nz > 0 && (rc = launch_zero_rows(a, stream))) return rc;
  if (nreg > 0 && (rc = launch_embed_reg(a, unit_fast, nreg, stream))) return rc;
  // num_lds_rows bounds the rows of FSW_REG_MAX_DEG < degree <= FSW_LDS_MAX_DEG: padded register path first, LDS path for the rest
  if (nlds > 0 && (rc = launch_embed_mid(a, unit_fast, nlds, stream))) return rc;
  if (nlds > 0 && (rc = launch_embed_lds(a, nlds, stream))) return rc;
  if (nglob > 0) {   // rows above FSW_LDS_MAX_DEG: hub kernels (unit weights, up to FSW_HUB_MAX_DEG), scratch-line kernel for the rest
    if (unit_fast) {
      if ((rc = launch_embed_hub(a, nglob, stream))) return rc;
      if ((rc = launch_embed_giant(a, nglob, stream))) return rc;
    } else if ((rc = launch_embed_global(a, nglob, stream))) {
      return rc;
    }
  }
  return 0;
}
