/* Tables handed to the Parquet decode kernels (host builds them from page/run HEADERS only; every value is
 * decoded on the device). */
#ifndef COMET_PARQUET_DEV_H
#define COMET_PARQUET_DEV_H
#include <stdint.h>

typedef struct PqPage {        /* one data page of a column (all selected row groups of a column are decoded by one launch) */
  int64_t row_start;           /* first row of the page within the column (flat columns: values == rows) */
  int64_t values_off;          /* staged-byte offset of the PLAIN values, or of the hybrid-encoded dictionary indices */
  int64_t str_first;           /* PLAIN BYTE_ARRAY pages: index of the page's first value in the string offset table */
  int32_t num_values;
  int32_t encoding;            /* 0 PLAIN, 1 dictionary indices (RLE/bit-packed hybrid) */
  int32_t bit_width;           /* dictionary index width */
  int32_t def_run_first, def_run_count;   /* definition-level runs (count 0: every value valid) */
  int32_t idx_run_first, idx_run_count;   /* dictionary-index runs */
  int32_t dict_offs_first;     /* strings: index of this page's dictionary in the offsets table */
  int64_t dict_off;            /* byte offset of this page's dictionary (its row group's) in the dictionary buffer */
  /* conversion of THIS page's values: files of one partition may store a column with different physical types (schema
   * evolution), and a column a file lacks is a synthetic page of NULLs or of its default value */
  int32_t kind;                /* PQ_* conversion */
  int32_t width;               /* source value width in bytes (FLBA length, 4, 8, 12) */
  int32_t dec_scale_up;        /* PQ_*_TO_DEC: multiply the unscaled value by 10^dec_scale_up (decimal scale widening) */
  /* page-index pruning keeps only some rows of a page: an entry then covers the kept rows [row_start, row_start + num_values) of the
   * OUTPUT, which are the page's rows lvl_skip … — the page's first lvl_skip levels and first val_skip (non-NULL) values are passed over */
  int32_t lvl_skip;
  int32_t val_skip;
  int32_t value_count;         /* non-NULL values of the page (= num_values for a page without NULLs): what its runs / PLAIN bytes hold */
  int32_t rep_run_first, rep_run_count;   /* repetition-level runs of a list leaf (count 0: every level 0) */
} PqPage;

typedef struct PqRun {         /* one run of an RLE / bit-packed hybrid section */
  int64_t byte_off;            /* bit-packed: staged-byte offset of the run's first group; PLAIN chunk: of its first value */
  int32_t value_start;         /* index of the run's first value within its page section */
  int32_t count;
  int32_t is_rle;              /* 0 bit-packed, 1 RLE, 2 = a chunk of a PLAIN page's values (index runs only: the unit of work of the
                                  run-at-a-time decode kernel; the row-at-a-time kernel never looks at the runs of a PLAIN page) */
  uint32_t rle_value;
  int32_t page;                /* index runs: the page (column-global index) the run belongs to */
  int32_t pad;
} PqRun;

typedef struct PqPendingRuns { /* a dictionary-encoded page whose index section the DEVICE walks (device/pq_runs.hpp): its run headers are not known to the host */
  int64_t begin, end;          /* the section, byte offsets in the column's byte buffer (the device-decompressed region) */
  int32_t bit_width;
  int32_t max_values;          /* stop behind this many values; -1: walk to the section's end */
  int32_t page;                /* column-global index of the page (pq_write_runs_kernel fills its idx_run_first / idx_run_count) */
  int32_t pad;
} PqPendingRuns;

typedef struct PqInflate {     /* one compressed page body the device decompresses (snappy_kernels.hip); offsets relative to the column's byte buffer */
  int64_t src_off;             /* compressed bytes, 16-byte aligned, readable up to the next multiple of 16 */
  int64_t dst_off;             /* where the decompressed page goes, 16-byte aligned */
  int32_t src_len, dst_len;
  int32_t preamble;            /* length of the stream's varint preamble (the host has seen the compressed bytes): where its first element starts */
  int32_t pad;
} PqInflate;

typedef struct PqCopyDesc {    /* one slice the upload kernel pulls from pinned host memory into HBM (both 16-byte aligned, len a multiple of 16) */
  const uint8_t* src;
  uint8_t* dst;
  uint64_t len;
} PqCopyDesc;

/* output conversions */
enum { PQ_COPY4 = 0, PQ_COPY8 = 1, PQ_I32_TO_I64 = 2, PQ_I32_TO_DEC = 3, PQ_I64_TO_DEC = 4, PQ_FLBA_TO_DEC = 5, PQ_BOOL = 6,
       PQ_I32_TO_I16 = 7, PQ_I32_TO_I8 = 8,
       /* schema adaptation (parquet/schema_adapter.rs, parquet_support.rs:141-240) */
       PQ_F32_TO_F64 = 9, PQ_I32_TO_F64 = 10, PQ_INT96_TO_TS_MICROS = 11,
       /* logical-type conversions: TIMESTAMP(MILLIS) → µs, UINT_32 → int64, UINT_64 → decimal(20,0) */
       PQ_I64_MILLIS_TO_MICROS = 12, PQ_U32_TO_I64 = 13, PQ_U64_TO_DEC = 14,
       /* RLE-encoded BOOLEAN pages (data page v2 writers): runs of 1-bit indices into the two-entry table {0, 1} */
       PQ_COPY1 = 15 };

typedef struct PqDecodeArgs {
  const PqPage* pages;
  int32_t npages;
  int32_t max_def;             /* 0 = required column */
  const PqRun* def_runs;
  const PqRun* idx_runs;
  const uint8_t* bytes;        /* staged, decompressed page bodies of the column chunk */
  const uint8_t* dict;         /* PLAIN dictionary values (fixed width) or dictionary string bytes */
  const int32_t* dict_offs;    /* strings: dictionary offsets (n_dict + 1) */
  const int64_t* plain_str_offs; /* PLAIN BYTE_ARRAY data pages: staged-byte offset of each value's bytes (+1 sentinel per page) */
  int64_t n_rows;              /* rows of this column chunk */
  int32_t out_width;           /* bytes per output value (fixed-width columns) */
  int32_t n_idx_runs;          /* entries of idx_runs (run-at-a-time kernel) */
  uint8_t* valid_out;          /* per-row validity bytes (at the chunk's row offset) */
  uint32_t* vidx;              /* per-row exclusive count of non-null rows before it (scratch) */
  void* values_out;            /* typed output at the chunk's row offset */
  uint32_t* lengths_out;       /* strings: per-row byte length */
  const int32_t* str_offsets;  /* strings (copy phase): output offsets at the chunk's row offset */
  uint8_t* str_bytes_out;      /* strings (copy phase): output data buffer */
  void* dense_out;             /* run-at-a-time kernel on a column with NULLs: values go here densely (value ordinal), expanded to rows afterwards */
  const PqRun* rep_runs;       /* list leaves: repetition-level runs */
  int32_t max_rep;
  int32_t pad;
} PqDecodeArgs;

#endif
