// Option store of libdroid_hip (see options.h) + the two C entry points that expose it.
#include "common.h"
#include "options.h"
#include <string.h>

namespace dh {

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

Options& opts() {
  static Options o = [] {
    Options v{};
    v.debug = env_int("DH_DEBUG", 0);
    v.chol_lookahead = env_int("DH_CHOL_LOOKAHEAD", 1);
    v.chol_regpanel = env_int("DH_CHOL_REGPANEL", 1);
    v.gram_strips = env_int("DH_GRAM_STRIPS", 0);
    v.conv_epi_staged = env_int("DH_CONV_EPI_STAGED", 1);
    v.conv_halo = env_int("DH_CONV_HALO", 1);
    v.conv_halo2 = env_int("DH_CONV_HALO2", 1);
    v.conv_dma = env_int("DH_CONV_DMA", 0);
    v.conv_xcd = env_int("DH_CONV_XCD", 1);
    v.dma_var = env_int("DH_DMA_VAR", 0);
    const char* pb = getenv("DH_PYR_BUILD");
    v.pyr_build_chunk = (pb && !strcmp(pb, "chunk")) ? 1 : 0;
    v.ba_strict = env_int("DH_BA_STRICT", 1);
    v.pyr_build_waves = env_int("DH_PYR_BUILD_WAVES", 8);
    v.pyr_build_tm = env_int("DH_PYR_BUILD_TM", 0);
    v.pyr_build_xcd = env_int("DH_PYR_BUILD_XCD", 0);
    v.pyr_build_dual = env_int("DH_PYR_BUILD_DUAL", 1);
    v.pyr_lds_pad = env_int("DH_PYR_LDS_PAD", 0);
    v.conv_nt_out = env_int("DH_CONV_NT_OUT", 1);
    v.lookup_mode = env_int("DH_LOOKUP_MODE", 0);
    v.lookup_fused = env_int("DH_LOOKUP_FUSED", 1);
    v.lookup_mix = env_int("DH_LOOKUP_MIX", 1);
    v.lookup_fill = env_int("DH_LOOKUP_FILL", 0);
    v.altcorr_v1 = env_int("DH_ALTCORR_V1", 0);
    v.conv_wino = env_int("DH_CONV_WINO", 0);
    v.conv_abl = env_int("DH_CONV_ABL", 0);
    v.conv_halo3 = env_int("DH_CONV_HALO3", 0);
    v.conv_halo4 = env_int("DH_CONV_HALO4", 0);
    v.conv_halo64 = env_int("DH_CONV_HALO64", 1);
    v.cinit_tiled = env_int("DH_CINIT_TILED", 1);
    v.conv_gate64 = env_int("DH_CONV_GATE64", 0);
    v.glo_fused = env_int("DH_GLO_FUSED", 1);
    v.eta_fused = env_int("DH_ETA_FUSED", 1);
    v.conv_two_tiles = env_int("DH_CONV_TWO_TILES", 0);
    v.conv_c7_split = env_int("DH_CONV_C7_SPLIT", 0);
    v.conv_c7_pp = env_int("DH_CONV_C7_PP", 0);
    v.conv_c7_w16 = env_int("DH_CONV_C7_W16", 0);
    v.conv_k1_half = env_int("DH_CONV_K1_HALF", 0);
    v.conv_two_tiles_maxc = env_int("DH_CONV_TWO_TILES_MAXC", 128);
#ifndef DH_ABLATION
    // release build: the prototype / timing-ablation kernels are not compiled in; a stray environment variable cannot
    // select a variant that returns wrong results (lookup_mode 2-5) or does not exist
    if (v.lookup_mode != 1 && v.lookup_mode != 6) v.lookup_mode = 0;
    if (v.chol_lookahead != 0) v.chol_lookahead = 1;
    v.conv_dma = 0; v.dma_var = 0; v.pyr_build_chunk = 0; v.altcorr_v1 = 0; v.conv_wino = 0; v.conv_abl = 0; v.conv_halo3 = 0; v.conv_halo4 = 0; v.lookup_fill = 0;
#endif
    return v;
  }();
  return o;
}

// values a release build accepts for the switches whose other settings need -DDH_ABLATION
static bool allowed(const char* name, int value) {
#ifdef DH_ABLATION
  (void)name; (void)value;
  return true;
#else
  if (!strcmp(name, "chol_lookahead")) return value == 0 || value == 1;                 // 2: dataflow schedule, -DDH_ABLATION builds
  if (!strcmp(name, "lookup_mode")) return value == 0 || value == 1 || value == 6;      // 1: nt tap loads, 6: synchronous twin -- same results (2-5, 7: timing ablations)
  if (!strcmp(name, "conv_dma") || !strcmp(name, "dma_var") || !strcmp(name, "pyr_build_chunk") || !strcmp(name, "altcorr_v1") ||
      !strcmp(name, "conv_wino") || !strcmp(name, "conv_abl") || !strcmp(name, "conv_halo3") || !strcmp(name, "conv_halo4") || !strcmp(name, "lookup_fill"))
    return value == 0;
  return true;
#endif
}

static int* slot(const char* name) {
  Options& o = opts();
  if (!name) return nullptr;
  if (!strcmp(name, "debug")) return &o.debug;
  if (!strcmp(name, "chol_lookahead")) return &o.chol_lookahead;
  if (!strcmp(name, "chol_regpanel")) return &o.chol_regpanel;
  if (!strcmp(name, "gram_strips")) return &o.gram_strips;
  if (!strcmp(name, "conv_epi_staged")) return &o.conv_epi_staged;
  if (!strcmp(name, "conv_halo")) return &o.conv_halo;
  if (!strcmp(name, "conv_halo2")) return &o.conv_halo2;
  if (!strcmp(name, "conv_dma")) return &o.conv_dma;
  if (!strcmp(name, "conv_xcd")) return &o.conv_xcd;
  if (!strcmp(name, "dma_var")) return &o.dma_var;
  if (!strcmp(name, "pyr_build_chunk")) return &o.pyr_build_chunk;
  if (!strcmp(name, "ba_strict")) return &o.ba_strict;
  if (!strcmp(name, "pyr_build_waves")) return &o.pyr_build_waves;
  if (!strcmp(name, "pyr_build_tm")) return &o.pyr_build_tm;
  if (!strcmp(name, "pyr_build_xcd")) return &o.pyr_build_xcd;
  if (!strcmp(name, "pyr_build_dual")) return &o.pyr_build_dual;
  if (!strcmp(name, "pyr_lds_pad")) return &o.pyr_lds_pad;
  if (!strcmp(name, "conv_nt_out")) return &o.conv_nt_out;
  if (!strcmp(name, "lookup_mode")) return &o.lookup_mode;
  if (!strcmp(name, "lookup_fused")) return &o.lookup_fused;
  if (!strcmp(name, "lookup_mix")) return &o.lookup_mix;
  if (!strcmp(name, "lookup_fill")) return &o.lookup_fill;
  if (!strcmp(name, "altcorr_v1")) return &o.altcorr_v1;
  if (!strcmp(name, "conv_wino")) return &o.conv_wino;
  if (!strcmp(name, "conv_abl")) return &o.conv_abl;
  if (!strcmp(name, "conv_halo3")) return &o.conv_halo3;
  if (!strcmp(name, "conv_halo4")) return &o.conv_halo4;
  if (!strcmp(name, "conv_halo64")) return &o.conv_halo64;
  if (!strcmp(name, "cinit_tiled")) return &o.cinit_tiled;
  if (!strcmp(name, "conv_gate64")) return &o.conv_gate64;
  if (!strcmp(name, "glo_fused")) return &o.glo_fused;
  if (!strcmp(name, "eta_fused")) return &o.eta_fused;
  if (!strcmp(name, "conv_two_tiles")) return &o.conv_two_tiles;
  if (!strcmp(name, "conv_c7_split")) return &o.conv_c7_split;
  if (!strcmp(name, "conv_c7_pp")) return &o.conv_c7_pp;
  if (!strcmp(name, "conv_c7_w16")) return &o.conv_c7_w16;
  if (!strcmp(name, "conv_k1_half")) return &o.conv_k1_half;
  if (!strcmp(name, "conv_two_tiles_maxc")) return &o.conv_two_tiles_maxc;
  return nullptr;
}

}  // namespace dh

static int g_options_epoch = 0;

extern "C" int dh_set_option(const char* name, int value) {
  int* s = dh::slot(name);
  if (!s) return DH_ERR_ARG;
  if (!dh::allowed(name, value)) return DH_ERR_UNSUPPORTED;
  if (*s != value) ++g_options_epoch;
  *s = value;
  return DH_OK;
}

extern "C" int dh_options_epoch(void) { return g_options_epoch; }

extern "C" int dh_get_option(const char* name, int* value) {
  if (name && value && !strcmp(name, "ablation_build")) {       // read-only: 1 when compiled with -DDH_ABLATION
#ifdef DH_ABLATION
    *value = 1;
#else
    *value = 0;
#endif
    return DH_OK;
  }
  int* s = dh::slot(name);
  if (!s || !value) return DH_ERR_ARG;
  *value = *s;
  return DH_OK;
}
