// Arrow C Data / C Stream / C Device interface structs (the stable public ABI,
// https://arrow.apache.org/docs/format/CDataInterface.html).  This is the wire the JVM side
// already speaks: NativeUtil.scala:69-84 allocates ArrowArray/ArrowSchema per output column,
// CometNativeArrowSource.scala:67 exports inputs as ArrowArrayStream
// (reference consumer: native/core/src/execution/operators/scan.rs:114-165).
#pragma once
#include <cstdint>

extern "C" {

#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
#define ARROW_FLAG_DICTIONARY_ORDERED 1
#define ARROW_FLAG_NULLABLE 2
#define ARROW_FLAG_MAP_KEYS_SORTED 4

struct ArrowSchema {
  const char* format;
  const char* name;
  const char* metadata;
  int64_t flags;
  int64_t n_children;
  struct ArrowSchema** children;
  struct ArrowSchema* dictionary;
  void (*release)(struct ArrowSchema*);
  void* private_data;
};

struct ArrowArray {
  int64_t length;
  int64_t null_count;
  int64_t offset;
  int64_t n_buffers;
  int64_t n_children;
  const void** buffers;
  struct ArrowArray** children;
  struct ArrowArray* dictionary;
  void (*release)(struct ArrowArray*);
  void* private_data;
};
#endif

#ifndef ARROW_C_STREAM_INTERFACE
#define ARROW_C_STREAM_INTERFACE
struct ArrowArrayStream {
  int (*get_schema)(struct ArrowArrayStream*, struct ArrowSchema* out);
  int (*get_next)(struct ArrowArrayStream*, struct ArrowArray* out);
  const char* (*get_last_error)(struct ArrowArrayStream*);
  void (*release)(struct ArrowArrayStream*);
  void* private_data;
};
#endif

#ifndef ARROW_C_DEVICE_DATA_INTERFACE
#define ARROW_C_DEVICE_DATA_INTERFACE
typedef int32_t ArrowDeviceType;
#define ARROW_DEVICE_CPU 1
#define ARROW_DEVICE_ROCM 10
#define ARROW_DEVICE_ROCM_HOST 11

struct ArrowDeviceArray {
  struct ArrowArray array;
  int64_t device_id;
  ArrowDeviceType device_type;
  void* sync_event;
  int64_t reserved[3];
};
#endif

#ifndef ARROW_C_DEVICE_STREAM_INTERFACE
#define ARROW_C_DEVICE_STREAM_INTERFACE
struct ArrowDeviceArrayStream {
  ArrowDeviceType device_type;
  int (*get_schema)(struct ArrowDeviceArrayStream*, struct ArrowSchema* out);
  int (*get_next)(struct ArrowDeviceArrayStream*, struct ArrowDeviceArray* out);
  const char* (*get_last_error)(struct ArrowDeviceArrayStream*);
  void (*release)(struct ArrowDeviceArrayStream*);
  void* private_data;
};
#endif

}  // extern "C"
