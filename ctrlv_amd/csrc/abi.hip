// ABI basics of libctrlv_hip.so / libctrlv_hip_f16.so: version, build id, thread-local error text, per-device property cache.
#include <stdarg.h>
#include <stdlib.h>

#include <mutex>

#include "common.h"

#ifndef CTRLV_BUILD_ID
#define CTRLV_BUILD_ID "unstamped"
#endif

static thread_local char g_err[512] = "";
void ctrlv_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

const ctrlv_debug_t& ctrlv_debug() {
  static const ctrlv_debug_t dbg = [] {
    auto env = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
    ctrlv_debug_t d;
    d.w16 = env("CTRLV_W16", 1);
    d.splitk = env("CTRLV_SPLITK", 1);
    d.force_tile = env("CTRLV_GEMM_FORCE_TILE", 0);
    d.conv_halo = env("CTRLV_CONV_HALO", 1);
    d.gn_fused = env("CTRLV_GN_FUSED", 1);
    d.gn_cross = env("CTRLV_GN_CROSS", 1);
    d.gn_rev = env("CTRLV_GN_REV", 1);
    d.gn_rows = env("CTRLV_GN_ROWS", 256);
    if (d.gn_rows < 32 || d.gn_rows > 1024) d.gn_rows = 256;
    d.ln_rows = env("CTRLV_LN_ROWS", 1);
    d.ff_fused = env("CTRLV_FF_FUSED", 1);
    d.ff_ln = env("CTRLV_FF_LN", 0);
    d.pp_balanced = env("CTRLV_PP_BALANCED", 1);
    d.pp_cgrp = env("CTRLV_PP_CGRP", 0);
    d.attn_rows = env("CTRLV_ATTN_ROWS", 0);
    d.temporal_fused = env("CTRLV_TEMPORAL_FUSED", 1);
    d.wgrad_pp = env("CTRLV_WGRAD_PP", 1);
    d.wgrad_slabs = env("CTRLV_WGRAD_SLABS", 0);
    return d;
  }();
  return dbg;
}

int ctrlv_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= CTRLV_MAX_DEVICES) dev = 0;
  return dev;
}

int ctrlv_num_cu(int dev) {
  static int cu[CTRLV_MAX_DEVICES] = {};
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (cu[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cu[dev] = n;
  }
  return cu[dev];
}

extern "C" int ctrlv_last_error(char* buf, size_t n) {
  if (buf && n) {
    strncpy(buf, g_err, n - 1);
    buf[n - 1] = 0;
  }
  return (int)strlen(g_err);
}

extern "C" int ctrlv_abi_version(void) { return 20; }
extern "C" int ctrlv_elem_dtype(void) { return CTRLV_ELEM_DTYPE; }

// sha256 prefix of ctrlv_amd/csrc/* + include/*.h at build time (stamped by __graft_entry__.build()): the host layer
// refuses a library whose id differs from the sources next to it.
extern "C" int ctrlv_build_id(char* buf, size_t n) {
  // the "CTRLV_BUILD_ID=" marker lets the build script read the id from the file without dlopen()ing it
  static const char marker[] = "CTRLV_BUILD_ID=" CTRLV_BUILD_ID;
  const char* id = marker + 15;
  if (buf && n) {
    strncpy(buf, id, n - 1);
    buf[n - 1] = 0;
  }
  return (int)strlen(id);
}
