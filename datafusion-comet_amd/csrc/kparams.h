/* Kernel-argument block shared by host (exec.cpp) and device (device/comet_device.hpp).
 * Plain C layout, passed by value as the single kernel argument. */
#ifndef COMET_KPARAMS_H
#define COMET_KPARAMS_H

#define COMET_MAX_IN 24
#define COMET_MAX_OUT 48

/* The error / aux block behind out[2] (COMET_ERR_BYTES, zeroed per execution):
 *   [0, 16)     u32 flags[2], u64 group counter
 *   [16, 192)   u64 kernel-level aux words of the aggregates
 *   [192, 448)  the DETAIL of the first raised error that carries one: u64 site + 1, u64 lo (or a string's length), u64 hi, u64 spare, then
 *               COMET_ERR_DETAIL_STR_BYTES bytes of a string value — what the Spark error JSON needs to name the offending value
 *   [448, 512)  scratch words of the executor */
#define COMET_ERR_BYTES 512
#define COMET_ERR_AUX_WORDS 22
#define COMET_ERR_DETAIL_WORD 24
#define COMET_ERR_DETAIL_STR_BYTES 224

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
