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
    v.lookup_mode = env_int("DH_LOOKUP_MODE", 0);
    v.lookup_fused = env_int("DH_LOOKUP_FUSED", 1);
    v.altcorr_v1 = env_int("DH_ALTCORR_V1", 0);
    v.conv_wino = env_int("DH_CONV_WINO", 0);
    return v;
  }();
  return o;
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
  if (!strcmp(name, "lookup_mode")) return &o.lookup_mode;
  if (!strcmp(name, "lookup_fused")) return &o.lookup_fused;
  if (!strcmp(name, "altcorr_v1")) return &o.altcorr_v1;
  if (!strcmp(name, "conv_wino")) return &o.conv_wino;
  return nullptr;
}

}  // namespace dh

static int g_options_epoch = 0;

extern "C" int dh_set_option(const char* name, int value) {
  int* s = dh::slot(name);
  if (!s) return DH_ERR_ARG;
  if (*s != value) ++g_options_epoch;
  *s = value;
  return DH_OK;
}

extern "C" int dh_options_epoch(void) { return g_options_epoch; }

extern "C" int dh_get_option(const char* name, int* value) {
  int* s = dh::slot(name);
  if (!s || !value) return DH_ERR_ARG;
  *value = *s;
  return DH_OK;
}
