// The hash exchange, independent of where the buffers live and of how the bytes travel (reference placement:
// native/shuffle/src/partitioners/multi_partition.rs:265-457, comet_partitioning.rs:51-57).
//
//   run<Ops>(ops, transport, columns, rows, keys) →
//     1. partition ids (murmur3 seed 42 chained over the key columns → pmod(world))            Ops
//     2. partition_starts / partition_row_indices, stable inside every partition               Ops
//     3. the COUNT exchange: every rank publishes its world + 1 partition starts               Transport::allgather_i64
//        → what each rank receives from whom, and where (Split)
//     4. per buffer: take into partition order (Ops) → the slices move                         Transport::alltoallv
//        validity and Boolean values travel one byte per row; Utf8 / Binary as int32 lengths (row split) plus the value bytes with their
//        OWN split (a second count exchange of the per-partition byte totals); the receiver rebuilds its offsets with one prefix sum
//
// Ops is the memory space: HipOps (exchange.cpp — HBM buffers, the partition / take / scan / pack kernels; the product) or the host
// stand-in of the CPU-only multi-process tests (tests/exchange_host/ — plain loops over malloc'ed buffers, test infrastructure that is
// NOT part of libcomet.so).  Transport is the wire: RCCL send / recv groups, the in-process rendezvous (both in exchange.cpp), or TCP
// (exchange_tcp.hpp — host memory; HipOps stages through pinned buffers).  Everything between — the splits, the order of the
// collectives, the validity-on-any-rank agreement, the byte splits, the offset rebuild — is this one template, so a world-size-2 / -4
// run over TCP on a CPU-only box executes the same code the RCCL path executes between GPUs.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/comet_amd.h"

namespace comet {
namespace xchg {

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// who sends how many units (rows, or bytes of a Utf8 column) to whom, from this rank's point of view
struct Split {
  std::vector<int64_t> starts;     // world + 1: unit offsets of my send buffer in partition order
  std::vector<int64_t> send;       // units I send to rank p
  std::vector<int64_t> recv;       // units I receive from rank s
  std::vector<int64_t> roff;       // world + 1: where sender s's slice begins in my receive buffer (senders in rank order)
  std::vector<int64_t> peer_off;   // where my slice begins in sender s's send buffer (transports that pull)
  int64_t total = 0;               // units I receive
};

class Transport {
 public:
  virtual ~Transport() {}
  virtual int world() const = 0;
  virtual int rank() const = 0;
  // the buffers handed to alltoallv must be host-addressable (TCP); otherwise they are whatever Ops allocates (device memory)
  virtual bool host_memory() const = 0;
  // one rank and no wire: the slices are copied in place by Ops
  virtual bool self_only() const { return false; }
  // every rank contributes n int64 values (host array); all = world · n values, rank major.  Collective.
  virtual void allgather_i64(const int64_t* mine, int n, int64_t* all) = 0;
  // units of w bytes: my slice for rank p is [sp.starts[p], + sp.send[p]) of send_buf; sender s's slice lands at sp.roff[s].  Collective.
  virtual void alltoallv(const void* send_buf, void* recv_buf, int w, const Split& sp) = 0;
};

// the count exchange: collective — every rank calls it the same number of times, in the same order
inline Split make_split(Transport& t, const std::vector<int64_t>& my_starts) {
  const int world = t.world(), rank = t.rank();
  if ((int)my_starts.size() != world + 1) throw Error("exchange: internal: partition starts arity");
  Split sp;
  sp.starts = my_starts;
  sp.send.assign((size_t)world, 0);
  sp.recv.assign((size_t)world, 0);
  sp.peer_off.assign((size_t)world, 0);
  for (int p = 0; p < world; p++) sp.send[(size_t)p] = my_starts[(size_t)p + 1] - my_starts[(size_t)p];
  std::vector<int64_t> all((size_t)world * (size_t)(world + 1), 0);
  if (t.self_only()) {
    memcpy(all.data(), my_starts.data(), (size_t)(world + 1) * 8);
  } else {
    t.allgather_i64(my_starts.data(), world + 1, all.data());
  }
  for (int s = 0; s < world; s++) {
    const int64_t* st = all.data() + (size_t)s * (size_t)(world + 1);   // sender s's partition starts
    for (int p = 0; p < world; p++)
      if (st[p + 1] < st[p] || st[0] != 0) throw Error("exchange: rank " + std::to_string(s) + " published partition starts that are not ascending from 0");
    sp.peer_off[(size_t)s] = st[rank];
    sp.recv[(size_t)s] = st[rank + 1] - st[rank];
  }
  sp.roff.assign((size_t)world + 1, 0);
  for (int s = 0; s < world; s++) {
    sp.roff[(size_t)s] = sp.total;
    sp.total += sp.recv[(size_t)s];
  }
  sp.roff[(size_t)world] = sp.total;
  return sp;
}

constexpr int kUtf8Column = 0, kBoolColumn = -1;   // value_width of the two kinds that are not fixed-width byte columns
// spark_expression.DataType.DataTypeId (types.proto:43-114)
inline int value_width(int type_id) {
  switch (type_id) {
    case 0: return kBoolColumn;                       // BOOL
    case 7: case 8: return kUtf8Column;               // STRING, BYTES
    case 1: return 1;                                 // INT8
    case 2: return 2;                                 // INT16
    case 3: case 12: case 5: return 4;                // INT32, DATE, FLOAT
    case 4: case 9: case 11: case 6: return 8;        // INT64, TIMESTAMP, TIMESTAMP_NTZ, DOUBLE
    case 10: return 16;                               // DECIMAL
    default: throw Error("exchange: column type " + std::to_string(type_id) + " is not supported by the in-library exchange yet");
  }
}

template <class Ops>
struct Result {
  int64_t rows = 0;
  std::vector<std::unique_ptr<typename Ops::Buf>> values, validity, aux;   // validity[c] null ⇔ column arrives without a bitmap; aux[c]: Utf8 bytes
  std::vector<int64_t> aux_bytes;
};

template <class Ops>
void run(Ops& ops, Transport& t, int32_t n_cols, const CometExchangeColumn* cols, int64_t rows, const int32_t* key_cols, int32_t n_keys, Result<Ops>& res) {
  using Buf = typename Ops::Buf;
  const int world = t.world();
  if (rows < 0 || rows >= ((int64_t)1 << 31)) throw Error("exchange: row count must be below 2^31");
  res.values.resize((size_t)n_cols);
  res.validity.resize((size_t)n_cols);
  res.aux.resize((size_t)n_cols);
  res.aux_bytes.assign((size_t)n_cols, 0);
  std::vector<int> width((size_t)n_cols);
  for (int i = 0; i < n_cols; i++) width[(size_t)i] = value_width(cols[i].type_id);
  for (int k = 0; k < n_keys; k++)
    if (key_cols[k] < 0 || key_cols[k] >= n_cols) throw Error("exchange: key column index out of range");

  // 1. + 2. partition ids, then partition_starts / partition_row_indices (stable inside every partition)
  Buf hashes, pids, idx;
  const size_t rows1 = (size_t)(rows > 0 ? rows : 1);
  hashes.ensure(rows1 * 4 + 16);
  pids.ensure(rows1 * 4 + 16);
  idx.ensure(rows1 * 4 + 16);
  std::vector<int64_t> starts((size_t)world + 1, 0);
  if (rows > 0) {
    ops.fill_u32((uint32_t*)hashes.p, rows, 42u);
    for (int k = 0; k < n_keys; k++) ops.murmur3(cols[key_cols[k]], rows, (uint32_t*)hashes.p);
    ops.pmod((const uint32_t*)hashes.p, rows, world, (int32_t*)pids.p);
    ops.partition_indices((const int32_t*)pids.p, rows, world, starts.data(), (uint32_t*)idx.p);
    if (starts[0] != 0 || starts[(size_t)world] != rows) throw Error("exchange: internal: partition starts do not cover the rows");
  }
  // 3. the row split
  const Split R = make_split(t, starts);
  const int64_t n_out = R.total;
  res.rows = n_out;
  if (n_out >= ((int64_t)1 << 31)) throw Error("exchange: a rank would receive 2^31 rows or more");

  // 4. every buffer: take into partition order, then move the slices
  typename Ops::HostBuf hsend, hrecv;
  auto move = [&](const void* send_buf, void* recv_buf, int w, const Split& sp) {   // w bytes per unit
    if (t.self_only()) {
      if (sp.total) ops.copy(recv_buf, send_buf, (size_t)sp.total * (size_t)w);
    } else if (t.host_memory() && Ops::kDeviceMemory) {
      // a wire that moves host bytes: stage the send buffer out and the received slices back in (pinned memory)
      const size_t sb = (size_t)sp.starts[(size_t)world] * (size_t)w, rb = (size_t)sp.total * (size_t)w;
      hsend.ensure(sb + 16);
      hrecv.ensure(rb + 16);
      if (sb) ops.to_host(hsend.p, send_buf, sb);
      t.alltoallv(hsend.p, hrecv.p, w, sp);
      if (rb) ops.from_host(recv_buf, hrecv.p, rb);
    } else {
      ops.before_transport();
      t.alltoallv(send_buf, recv_buf, w, sp);
    }
  };
  // does the column carry validity on ANY rank?  (a rank without NULLs still has to send validity bytes then)
  std::vector<int64_t> has_valid((size_t)n_cols, 0);
  for (int i = 0; i < n_cols; i++) has_valid[(size_t)i] = cols[i].validity ? 1 : 0;
  if (!t.self_only() && n_cols > 0) {
    std::vector<int64_t> all((size_t)n_cols * (size_t)world, 0);
    t.allgather_i64(has_valid.data(), n_cols, all.data());
    for (int s = 0; s < world; s++)
      for (int i = 0; i < n_cols; i++) has_valid[(size_t)i] |= all[(size_t)s * (size_t)n_cols + (size_t)i] ? 1 : 0;
  }
  Buf send_buf, vbytes_send, vbytes_recv, lengths, send_offs, recv_lengths;
  const size_t out1 = (size_t)(n_out > 0 ? n_out : 1);
  for (int i = 0; i < n_cols; i++) {
    const int w = width[(size_t)i];
    res.values[(size_t)i].reset(new Buf());
    if (w > 0) {
      send_buf.ensure(rows1 * (size_t)w + 16);
      if (rows > 0) ops.take(w, cols[i].values, (const uint32_t*)idx.p, rows, send_buf.p);
      res.values[(size_t)i]->ensure(out1 * (size_t)w + 16);
      move(send_buf.p, res.values[(size_t)i]->p, w, R);
    } else if (w == kBoolColumn) {
      // bit-packed values: one byte per row on the wire (partition boundaries are not byte aligned), packed again on arrival
      vbytes_send.ensure(rows1 + 16);
      vbytes_recv.ensure(out1 + 16);
      if (rows > 0) ops.take_valid_bytes((const uint8_t*)cols[i].values, (const uint32_t*)idx.p, rows, (uint8_t*)vbytes_send.p);
      move(vbytes_send.p, vbytes_recv.p, 1, R);
      res.values[(size_t)i]->ensure((size_t)((n_out + 7) / 8) + 16);
      if (n_out > 0) ops.pack((const uint8_t*)vbytes_recv.p, (uint8_t*)res.values[(size_t)i]->p, n_out);
    } else {
      // Utf8 / Binary: lengths (0 for NULL rows) → offsets of my send bytes → the bytes in partition order
      const int32_t* offs = (const int32_t*)cols[i].values;
      lengths.ensure(rows1 * 4 + 16);
      send_offs.ensure((rows1 + 1) * 4 + 16);
      std::vector<int64_t> bstarts((size_t)world + 1, 0);
      if (rows > 0) {
        ops.take_utf8_lengths(offs, (const uint32_t*)idx.p, cols[i].validity, rows, (uint32_t*)lengths.p);
        ops.scan_u32((const uint32_t*)lengths.p, rows, (int32_t*)send_offs.p);
        std::vector<int32_t> at((size_t)world + 1, 0);
        ops.read_i32_at((const int32_t*)send_offs.p, starts.data(), world + 1, at.data());    // byte offset at every partition start
        for (int p = 0; p <= world; p++) bstarts[(size_t)p] = at[(size_t)p];
        // (`idx` is a permutation of the column's rows and the column's own offsets are 32-bit, so the lengths sum to less than 2^31 and the
        // 32-bit prefix sums cannot wrap; the partition starts are checked all the same — ADVICE r3)
        for (int p = 0; p < world; p++)
          if (bstarts[(size_t)p] < 0 || bstarts[(size_t)p + 1] < bstarts[(size_t)p]) throw Error("exchange: Utf8 column exceeds 2 GiB of string data (LargeUtf8 is not supported)");
      }
      send_buf.ensure((size_t)(bstarts[(size_t)world] > 0 ? bstarts[(size_t)world] : 1) + 16);
      if (rows > 0) ops.take_utf8_copy(offs, cols[i].aux, (const uint32_t*)idx.p, cols[i].validity, rows, (const int32_t*)send_offs.p, (uint8_t*)send_buf.p);
      const Split B = make_split(t, bstarts);        // the byte split: a second count exchange
      if (B.total >= ((int64_t)1 << 31)) throw Error("exchange: a rank would receive 2 GiB or more of one Utf8 column");
      recv_lengths.ensure(out1 * 4 + 16);
      move(lengths.p, recv_lengths.p, 4, R);
      res.aux[(size_t)i].reset(new Buf());
      res.aux[(size_t)i]->ensure((size_t)(B.total > 0 ? B.total : 1) + 16);
      move(send_buf.p, res.aux[(size_t)i]->p, 1, B);
      res.aux_bytes[(size_t)i] = B.total;
      // the received slices arrive sender after sender, each in row order: one prefix sum over the lengths is the offsets buffer
      res.values[(size_t)i]->ensure((out1 + 1) * 4 + 16);
      if (n_out > 0) ops.scan_u32((const uint32_t*)recv_lengths.p, n_out, (int32_t*)res.values[(size_t)i]->p);
      else ops.set_bytes(res.values[(size_t)i]->p, 0, 4);
    }
    if (has_valid[(size_t)i]) {
      vbytes_send.ensure(rows1 + 16);
      vbytes_recv.ensure(out1 + 16);
      if (rows > 0) {
        if (cols[i].validity) ops.take_valid_bytes(cols[i].validity, (const uint32_t*)idx.p, rows, (uint8_t*)vbytes_send.p);
        else ops.set_bytes(vbytes_send.p, 1, (size_t)rows);
      }
      move(vbytes_send.p, vbytes_recv.p, 1, R);
      res.validity[(size_t)i].reset(new Buf());
      res.validity[(size_t)i]->ensure((size_t)((n_out + 7) / 8) + 16);
      if (n_out > 0) ops.pack((const uint8_t*)vbytes_recv.p, (uint8_t*)res.validity[(size_t)i]->p, n_out);
    }
  }
  ops.sync();     // scratch buffers return to their pool; the result is complete
}

}  // namespace xchg
}  // namespace comet
