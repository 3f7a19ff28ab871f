/* Kernel-argument block shared by host (exec.cpp) and device (device/comet_device.hpp).
 * Plain C layout, passed by value as the single kernel argument. */
#ifndef COMET_KPARAMS_H
#define COMET_KPARAMS_H

#define COMET_MAX_IN 24
#define COMET_MAX_OUT 48

typedef struct CometCol {
  const void* data;            /* values buffer (or int32 offsets for Utf8) */
  const unsigned char* valid;  /* Arrow validity bitmap (LSB first) or NULL when null_count == 0 */
  const void* aux;             /* Utf8: data bytes */
  long long offset;            /* logical element offset into the buffers (Arrow `offset`) */
} CometCol;

typedef struct CometKParams {
  long long n;                 /* rows in this launch */
  long long iarg[7];           /* per-kernel integers (table capacity, partial count, ...) */
  CometCol in[COMET_MAX_IN];
  void* out[COMET_MAX_OUT];    /* outputs / scratch, meaning defined per kernel template */
} CometKParams;

#endif
