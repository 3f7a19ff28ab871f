// TCP transport of the hash exchange (exchange_core.hpp Transport): a full mesh of sockets between the ranks, one process per rank —
// for ranks that share no RCCL-capable fabric (another node, a CPU-only box), and the wire the multi-process CPU tests drive the
// exchange over.  Host memory only; HipOps stages through pinned buffers.
//
//   rendezvous  `peers` = "host:port,host:port,…", one entry per rank.  Rank r listens on its own port; for every pair a < b, b connects to
//               a and says who it is (magic, world, rank).  Connecting retries until the deadline: ranks start in any order.
//   progress    non-blocking sockets and one poll() loop per collective: every rank sends to and receives from all its peers at once, so
//               no ordering of sends and receives can deadlock whatever the slice sizes.
//   framing     every message carries (magic, sequence number, byte count); the receiver knows what to expect from the count exchange and
//               refuses anything else — two ranks that disagree about the collective they are in fail instead of mixing their bytes.
//   failure     a peer that closes its socket (its process died) or stays silent for `timeout_ms` fails the collective with an error
//               naming the peer; nothing hangs.
#pragma once
#include <arpa/inet.h>
#include <errno.h>
#include <fcntl.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <sys/types.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "exchange_core.hpp"

namespace comet {
namespace xchg {

class TcpTransport : public Transport {
 public:
  TcpTransport(const std::string& peers, int world, int rank, int timeout_ms) : world_(world), rank_(rank), timeout_ms_(timeout_ms > 0 ? timeout_ms : 60000) {
    if (world < 1 || rank < 0 || rank >= world) throw Error("exchange: bad rank / world");
    std::vector<std::pair<std::string, int>> addr;
    size_t pos = 0;
    while (pos <= peers.size()) {
      size_t e = peers.find(',', pos);
      if (e == std::string::npos) e = peers.size();
      std::string item = peers.substr(pos, e - pos);
      size_t c = item.rfind(':');
      if (c == std::string::npos || c == 0 || c + 1 >= item.size()) throw Error("exchange: tcp peer '" + item + "' is not host:port");
      addr.emplace_back(item.substr(0, c), atoi(item.c_str() + c + 1));
      pos = e + 1;
    }
    if ((int)addr.size() != world) throw Error("exchange: tcp peer list holds " + std::to_string(addr.size()) + " entries for " + std::to_string(world) + " ranks");
    fds_.assign((size_t)world, -1);
    try {
      rendezvous(addr);
    } catch (...) {
      close_all();
      throw;
    }
  }
  ~TcpTransport() override { close_all(); }
  int world() const override { return world_; }
  int rank() const override { return rank_; }
  bool host_memory() const override { return true; }
  bool self_only() const override { return world_ == 1; }

  void allgather_i64(const int64_t* mine, int n, int64_t* all) override {
    std::vector<Leg> send((size_t)world_), recv((size_t)world_);
    for (int p = 0; p < world_; p++) {
      send[(size_t)p] = Leg{(char*)mine, (size_t)n * 8};
      recv[(size_t)p] = Leg{(char*)(all + (size_t)p * (size_t)n), (size_t)n * 8};
    }
    memcpy(all + (size_t)rank_ * (size_t)n, mine, (size_t)n * 8);
    collective(send, recv, "count exchange");
  }
  void alltoallv(const void* send_buf, void* recv_buf, int w, const Split& sp) override {
    std::vector<Leg> send((size_t)world_), recv((size_t)world_);
    for (int p = 0; p < world_; p++) {
      send[(size_t)p] = Leg{(char*)send_buf + (size_t)sp.starts[(size_t)p] * (size_t)w, (size_t)sp.send[(size_t)p] * (size_t)w};
      recv[(size_t)p] = Leg{(char*)recv_buf + (size_t)sp.roff[(size_t)p] * (size_t)w, (size_t)sp.recv[(size_t)p] * (size_t)w};
    }
    // my own slice never touches a socket
    if (recv[(size_t)rank_].n) memcpy(recv[(size_t)rank_].p, send[(size_t)rank_].p, recv[(size_t)rank_].n);
    collective(send, recv, "slice exchange");
  }

 private:
  struct Leg { char* p; size_t n; };
  struct Header { uint32_t magic; uint32_t seq; uint64_t bytes; };
  static constexpr uint32_t kMagic = 0x43584d54u;   // "TMXC"
  int world_, rank_, timeout_ms_;
  std::vector<int> fds_;
  uint32_t seq_ = 0;

  static int64_t now_ms() { return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  void close_all() {
    for (int& fd : fds_) {
      if (fd >= 0) ::close(fd);
      fd = -1;
    }
  }
  static void set_nonblocking(int fd) {
    int fl = fcntl(fd, F_GETFL, 0);
    fcntl(fd, F_SETFL, fl | O_NONBLOCK);
    int one = 1;
    setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
  }
  static void write_fully(int fd, const void* p, size_t n, int64_t deadline) {
    const char* c = (const char*)p;
    while (n) {
      ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL);
      if (k > 0) { c += k; n -= (size_t)k; continue; }
      if (k < 0 && (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR)) {
        if (now_ms() > deadline) throw Error("exchange: tcp rendezvous timed out");
        struct pollfd pf = {fd, POLLOUT, 0};
        poll(&pf, 1, 50);
        continue;
      }
      throw Error(std::string("exchange: tcp rendezvous send failed: ") + strerror(errno));
    }
  }
  static void read_fully(int fd, void* p, size_t n, int64_t deadline) {
    char* c = (char*)p;
    while (n) {
      ssize_t k = ::recv(fd, c, n, 0);
      if (k > 0) { c += k; n -= (size_t)k; continue; }
      if (k == 0) throw Error("exchange: tcp rendezvous: the peer closed the connection");
      if (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR) {
        if (now_ms() > deadline) throw Error("exchange: tcp rendezvous timed out");
        struct pollfd pf = {fd, POLLIN, 0};
        poll(&pf, 1, 50);
        continue;
      }
      throw Error(std::string("exchange: tcp rendezvous receive failed: ") + strerror(errno));
    }
  }
  void rendezvous(const std::vector<std::pair<std::string, int>>& addr) {
    const int64_t deadline = now_ms() + timeout_ms_;
    int lfd = -1;
    const int n_accept = world_ - 1 - rank_;          // ranks above me connect to me
    if (n_accept > 0) {
      lfd = ::socket(AF_INET, SOCK_STREAM, 0);
      if (lfd < 0) throw Error(std::string("exchange: socket: ") + strerror(errno));
      int one = 1;
      setsockopt(lfd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
      struct sockaddr_in sa;
      memset(&sa, 0, sizeof sa);
      sa.sin_family = AF_INET;
      sa.sin_addr.s_addr = htonl(INADDR_ANY);
      sa.sin_port = htons((uint16_t)addr[(size_t)rank_].second);
      if (::bind(lfd, (struct sockaddr*)&sa, sizeof sa) != 0 || ::listen(lfd, world_) != 0) {
        const std::string why = strerror(errno);
        ::close(lfd);
        throw Error("exchange: cannot listen on port " + std::to_string(addr[(size_t)rank_].second) + ": " + why);
      }
      set_nonblocking(lfd);
    }
    struct Guard { int fd; ~Guard() { if (fd >= 0) ::close(fd); } } guard{lfd};
    // connect to every rank below me (they may not be listening yet: retry until the deadline)
    for (int p = 0; p < rank_; p++) {
      struct addrinfo hints, *ai = nullptr;
      memset(&hints, 0, sizeof hints);
      hints.ai_family = AF_INET;
      hints.ai_socktype = SOCK_STREAM;
      if (getaddrinfo(addr[(size_t)p].first.c_str(), std::to_string(addr[(size_t)p].second).c_str(), &hints, &ai) != 0 || !ai)
        throw Error("exchange: cannot resolve tcp peer " + addr[(size_t)p].first);
      int fd = -1;
      for (;;) {
        fd = ::socket(AF_INET, SOCK_STREAM, 0);
        if (fd >= 0 && ::connect(fd, ai->ai_addr, ai->ai_addrlen) == 0) break;
        if (fd >= 0) ::close(fd);
        fd = -1;
        if (now_ms() > deadline) break;
        usleep(20000);
      }
      freeaddrinfo(ai);
      if (fd < 0) throw Error("exchange: rank " + std::to_string(p) + " (" + addr[(size_t)p].first + ":" + std::to_string(addr[(size_t)p].second) + ") did not accept a connection within " + std::to_string(timeout_ms_) + " ms");
      set_nonblocking(fd);
      fds_[(size_t)p] = fd;
      const uint32_t hello[3] = {kMagic, (uint32_t)world_, (uint32_t)rank_};
      write_fully(fd, hello, sizeof hello, deadline);
    }
    for (int got = 0; got < n_accept;) {
      struct pollfd pf = {lfd, POLLIN, 0};
      poll(&pf, 1, 50);
      int fd = ::accept(lfd, nullptr, nullptr);
      if (fd < 0) {
        if (now_ms() > deadline) throw Error("exchange: only " + std::to_string(got) + " of " + std::to_string(n_accept) + " higher ranks connected within " + std::to_string(timeout_ms_) + " ms");
        continue;
      }
      set_nonblocking(fd);
      uint32_t hello[3] = {0, 0, 0};
      // (a connection that is not one of ours — a port scanner, a health check — is dropped, it does not end the rendezvous: ADVICE r3.  This
      // wire is the CPU suite's test vehicle; ranks of a job find each other through RCCL)
      try {
        read_fully(fd, hello, sizeof hello, std::min<int64_t>(deadline, now_ms() + 2000));
      } catch (...) {
        ::close(fd);
        continue;
      }
      if (hello[0] != kMagic || (int)hello[1] != world_ || (int)hello[2] <= rank_ || (int)hello[2] >= world_ || fds_[hello[2]] >= 0) {
        ::close(fd);
        continue;
      }
      fds_[hello[2]] = fd;
      got++;
    }
  }

  // every peer: send one framed message, receive one framed message — all at once
  void collective(const std::vector<Leg>& send, const std::vector<Leg>& recv, const char* what) {
    const uint32_t seq = ++seq_;
    struct Peer { Header sh, rh; size_t s_done = 0, r_done = 0; bool s_fin = false, r_fin = false; };
    std::vector<Peer> st((size_t)world_);
    int open = 0;
    for (int p = 0; p < world_; p++) {
      if (p == rank_) { st[(size_t)p].s_fin = st[(size_t)p].r_fin = true; continue; }
      st[(size_t)p].sh = Header{kMagic, seq, (uint64_t)send[(size_t)p].n};
      open += 2;
    }
    int64_t last_progress = now_ms();
    std::vector<struct pollfd> pfs;
    std::vector<int> who;
    while (open > 0) {
      pfs.clear();
      who.clear();
      for (int p = 0; p < world_; p++) {
        Peer& s = st[(size_t)p];
        if (s.s_fin && s.r_fin) continue;
        struct pollfd pf = {fds_[(size_t)p], (short)((s.s_fin ? 0 : POLLOUT) | (s.r_fin ? 0 : POLLIN)), 0};
        pfs.push_back(pf);
        who.push_back(p);
      }
      const int rc = poll(pfs.data(), (nfds_t)pfs.size(), 100);
      if (rc < 0 && errno != EINTR) throw Error(std::string("exchange: poll: ") + strerror(errno));
      bool progressed = false;
      for (size_t k = 0; k < pfs.size(); k++) {
        const int p = who[k];
        Peer& s = st[(size_t)p];
        const int fd = fds_[(size_t)p];
        if (!s.s_fin && (pfs[k].revents & (POLLOUT | POLLERR | POLLHUP))) {
          const size_t total = sizeof(Header) + send[(size_t)p].n;
          while (s.s_done < total) {
            const char* src;
            size_t n;
            if (s.s_done < sizeof(Header)) { src = (const char*)&s.sh + s.s_done; n = sizeof(Header) - s.s_done; }
            else { src = send[(size_t)p].p + (s.s_done - sizeof(Header)); n = total - s.s_done; }
            const ssize_t w = ::send(fd, src, n, MSG_NOSIGNAL);
            if (w > 0) { s.s_done += (size_t)w; progressed = true; continue; }
            if (w < 0 && (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR)) break;
            throw Error(std::string("exchange: ") + what + ": rank " + std::to_string(p) + " is gone (send: " + strerror(errno) + ")");
          }
          if (s.s_done == total) { s.s_fin = true; open--; }
        }
        if (!s.r_fin && (pfs[k].revents & (POLLIN | POLLERR | POLLHUP))) {
          for (;;) {
            char* dst;
            size_t n;
            if (s.r_done < sizeof(Header)) { dst = (char*)&s.rh + s.r_done; n = sizeof(Header) - s.r_done; }
            else {
              const size_t total = sizeof(Header) + recv[(size_t)p].n;
              if (s.r_done == total) break;
              dst = recv[(size_t)p].p + (s.r_done - sizeof(Header));
              n = total - s.r_done;
            }
            const ssize_t r = ::recv(fd, dst, n, 0);
            if (r > 0) {
              s.r_done += (size_t)r;
              progressed = true;
              if (s.r_done == sizeof(Header)) {
                if (s.rh.magic != kMagic || s.rh.seq != seq || s.rh.bytes != (uint64_t)recv[(size_t)p].n)
                  throw Error(std::string("exchange: ") + what + ": rank " + std::to_string(p) + " sent message " + std::to_string(s.rh.seq) + " of " + std::to_string(s.rh.bytes) +
                              " bytes where message " + std::to_string(seq) + " of " + std::to_string(recv[(size_t)p].n) + " bytes was expected (the ranks disagree about the exchange)");
              }
              continue;
            }
            if (r == 0) throw Error(std::string("exchange: ") + what + ": rank " + std::to_string(p) + " closed its connection (its process ended mid-exchange)");
            if (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR) break;
            throw Error(std::string("exchange: ") + what + ": rank " + std::to_string(p) + " is gone (receive: " + strerror(errno) + ")");
          }
          if (s.r_done == sizeof(Header) + recv[(size_t)p].n) { s.r_fin = true; open--; }
        }
      }
      const int64_t t = now_ms();
      if (progressed) last_progress = t;
      else if (t - last_progress > timeout_ms_) {
        std::string silent;
        for (int p = 0; p < world_; p++)
          if (!st[(size_t)p].r_fin) silent += (silent.empty() ? "" : ", ") + std::to_string(p);
        throw Error(std::string("exchange: ") + what + ": no byte moved for " + std::to_string(timeout_ms_) + " ms (waiting for rank " + silent + ")");
      }
    }
  }
};

}  // namespace xchg
}  // namespace comet
