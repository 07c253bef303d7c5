// Replay simulation of the dense unprojection's feature-line traffic through the 8 private L2s (+ a shared Infinity
// Cache) of an MI355X, for one 32-channel sweep at the north-star shape (V=40, 480x640 maps, 192^3 grid): which
// voxel -> workgroup -> XCD schedules keep the re-reads along camera rays on chip, and what bound is left.
//
//   g++ -O3 -fopenmp -o /tmp/l2sim scripts/dense_l2sim.cpp && /tmp/l2sim <schedule> [key=value ...]
//
// Model: a workgroup = 256 voxels; per view it requests one 128-B line per valid voxel (pixel x 32 channels); an XCD
// runs R workgroups at once, each advancing view by view at a cost of (c0 + lines requested) time units (event
// driven: workgroups with few valid voxels run ahead, like on the chip); a finished workgroup is replaced by the next
// block the round-robin dispatcher hands that XCD.  L2: 4 MiB, 128-B lines, 16-way LRU.  Infinity Cache: 256 MiB,
// 16-way LRU over the merged miss stream.  Prints hit rates and the implied fabric / HBM read bytes for 8 sweeps.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <queue>
#include <string>
#include <vector>

static int V = 40, H = 480, W = 640, X = 192, Y = 192, Z = 192;
static const float VS = 0.04f;

struct Cache {
  int sets, ways;
  std::vector<uint32_t> tag, age;
  uint32_t clock = 0;
  uint64_t hit = 0, miss = 0;
  Cache(size_t bytes, int ways_) : ways(ways_) {
    sets = (int)(bytes / 128 / ways_);
    tag.assign((size_t)sets * ways, 0xffffffffu);
    age.assign((size_t)sets * ways, 0);
  }
  bool access(uint32_t line) {
    uint32_t h = line * 2654435761u;
    int s = (int)((h >> 8) % (uint32_t)sets);
    uint32_t* t = &tag[(size_t)s * ways];
    uint32_t* a = &age[(size_t)s * ways];
    ++clock;
    int lru = 0;
    for (int w = 0; w < ways; ++w) {
      if (t[w] == line) { a[w] = clock; ++hit; return true; }
      if (a[w] < a[lru]) lru = w;
    }
    t[lru] = line; a[lru] = clock; ++miss;
    return false;
  }
};

static std::vector<int32_t> pix;   // [V][G]
static int64_t G;

static void make_pix() {
  G = (int64_t)X * Y * Z;
  pix.resize((size_t)V * G);
  double ext[3] = {X * 0.04, Y * 0.04, Z * 0.04};
  double f = 577.0 * W / 1296.0;
  for (int v = 0; v < V; ++v) {
    double a = 2 * M_PI * v / V;
    double eye[3] = {ext[0] / 2 + 1.2 * cos(a), ext[1] / 2 + 1.2 * sin(a), ext[2] / 2 + 0.2};
    double tgt[3] = {ext[0] / 2 + 3 * cos(a + 2.5), ext[1] / 2 + 3 * sin(a + 2.5), ext[2] / 2 - 0.3};
    double fw[3] = {tgt[0] - eye[0], tgt[1] - eye[1], tgt[2] - eye[2]};
    double n = sqrt(fw[0] * fw[0] + fw[1] * fw[1] + fw[2] * fw[2]);
    for (double& q : fw) q /= n;
    double rt[3] = {fw[1] * 1 - fw[2] * 0, fw[2] * 0 - fw[0] * 1, 0};
    n = sqrt(rt[0] * rt[0] + rt[1] * rt[1]);
    rt[0] /= n; rt[1] /= n;
    double dn[3] = {fw[1] * rt[2] - fw[2] * rt[1], fw[2] * rt[0] - fw[0] * rt[2], fw[0] * rt[1] - fw[1] * rt[0]};
    double R[3][3] = {{rt[0], rt[1], rt[2]}, {dn[0], dn[1], dn[2]}, {fw[0], fw[1], fw[2]}};
    double t[3];
    for (int r = 0; r < 3; ++r) t[r] = -(R[r][0] * eye[0] + R[r][1] * eye[1] + R[r][2] * eye[2]);
    double K[3][3] = {{f, 0, W / 2.0}, {0, f, H / 2.0}, {0, 0, 1}};
    float P[3][4];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += K[r][k] * (c < 3 ? R[k][c] : t[k]);
        P[r][c] = (float)s;
      }
#pragma omp parallel for
    for (int x = 0; x < X; ++x)
      for (int y = 0; y < Y; ++y)
        for (int z = 0; z < Z; ++z) {
          float wx = x * VS, wy = y * VS, wz = z * VS;
          float cam[3];
          for (int r = 0; r < 3; ++r) cam[r] = P[r][0] * wx + P[r][1] * wy + P[r][2] * wz + P[r][3];
          float rx = rintf(cam[0] / cam[2]), ry = rintf(cam[1] / cam[2]);
          bool ok = rx >= 0 && ry >= 0 && rx < W && ry < H && cam[2] > 0;
          pix[(size_t)v * G + ((int64_t)x * Y + y) * Z + z] = ok ? (int)ry * W + (int)rx : -1;
        }
  }
}

// ---- schedules: logical order of voxels (blocks of `wg` consecutive entries = one workgroup) + block -> XCD queues
struct Sched {
  std::vector<int32_t> vox;            // voxel per (block, thread), -1 = padding
  std::vector<std::vector<int>> queue; // per XCD: block ids in dispatch order
  int wg = 256;
};

static int64_t lin(int x, int y, int z) { return ((int64_t)x * Y + y) * Z + z; }

// bricks of sx*sy*sz voxels, inside a brick: tiles of tt x tt columns, z fastest inside a column (run zi)
static void brick_voxels(std::vector<int32_t>& out, int bx, int by, int bz, int sx, int sy, int sz, int tt) {
  for (int tx = 0; tx < sx; tx += tt)
    for (int ty = 0; ty < sy; ty += tt)
      for (int cx = 0; cx < tt; ++cx)
        for (int cy = 0; cy < tt; ++cy)
          for (int z = 0; z < sz; ++z) {
            int x = bx * sx + tx + cx, y = by * sy + ty + cy, zz = bz * sz + z;
            out.push_back(x < X && y < Y && zz < Z ? (int32_t)lin(x, y, zz) : -1);
          }
}

static uint32_t morton3(uint32_t x, uint32_t y, uint32_t z) {
  auto sp = [](uint32_t v) { uint32_t r = 0; for (int i = 0; i < 10; ++i) r |= ((v >> i) & 1u) << (3 * i); return r; };
  return sp(x) | (sp(y) << 1) | (sp(z) << 2);
}

int main(int argc, char** argv) {
  std::string sched = argc > 1 ? argv[1] : "brick";
  int R = 192, sx = 16, sy = 16, sz = 32, tt = 8, wg = 256, lockstep = 0, nx = 8, super = 0, interleave = 0;
  double c0 = 16, l2mb = 4.0, mallmb = 256;
  std::string border = "yxz";
  for (int i = 2; i < argc; ++i) {
    std::string a = argv[i];
    auto eq = a.find('=');
    std::string k = a.substr(0, eq), val = a.substr(eq + 1);
    if (k == "R") R = atoi(val.c_str());
    else if (k == "sx") sx = atoi(val.c_str());
    else if (k == "sy") sy = atoi(val.c_str());
    else if (k == "sz") sz = atoi(val.c_str());
    else if (k == "tt") tt = atoi(val.c_str());
    else if (k == "wg") wg = atoi(val.c_str());
    else if (k == "c0") c0 = atof(val.c_str());
    else if (k == "l2mb") l2mb = atof(val.c_str());
    else if (k == "mallmb") mallmb = atof(val.c_str());
    else if (k == "lockstep") lockstep = atoi(val.c_str());
    else if (k == "nx") nx = atoi(val.c_str());
    else if (k == "order") border = val;
    else if (k == "super") super = atoi(val.c_str());
    else if (k == "V") V = atoi(val.c_str());
    else if (k == "interleave") interleave = atoi(val.c_str());
    else { fprintf(stderr, "unknown key %s\n", k.c_str()); return 1; }
  }
  make_pix();
  Sched S;
  S.wg = wg;
  S.queue.resize(nx);
  if (sched == "linear") {
    // z-fastest linear order, x-plane chunks per XCD group (round-1 kernel)
    for (int64_t g = 0; g < G; ++g) S.vox.push_back((int32_t)g);
    int64_t nb = (G + wg - 1) / wg, cb = std::max<int64_t>(32, ((int64_t)Y * Z + wg - 1) / wg);
    for (int64_t c = 0; c * cb < nb; ++c)
      for (int64_t k = 0; k < cb && c * cb + k < nb; ++k) S.queue[c % nx].push_back((int)(c * cb + k));
  } else {
    // bricks; order of bricks: "yxz" (y fastest, then x, then z: the round-2 kernel), "morton", "zyx" ...
    int nbx = (X + sx - 1) / sx, nby = (Y + sy - 1) / sy, nbz = (Z + sz - 1) / sz;
    struct B { int x, y, z; uint32_t key; };
    std::vector<B> bricks;
    for (int bz = 0; bz < nbz; ++bz)
      for (int bx = 0; bx < nbx; ++bx)
        for (int by = 0; by < nby; ++by) {
          uint32_t key;
          if (border == "morton") key = morton3(bx, by, bz);
          else if (border == "zfast") key = (bx * nby + by) * nbz + bz;
          else key = (bz * nbx + bx) * nby + by;
          bricks.push_back({bx, by, bz, key});
        }
    std::stable_sort(bricks.begin(), bricks.end(), [](const B& a, const B& b) { return a.key < b.key; });
    int per = sx * sy * sz / wg;
    for (size_t i = 0; i < bricks.size(); ++i) {
      brick_voxels(S.vox, bricks[i].x, bricks[i].y, bricks[i].z, sx, sy, sz, tt);
      // "brick": chunk i -> XCD i % nx (current kernel).  "own": every XCD walks ALL bricks (sweep per XCD): simulate one
      // XCD with the full queue.  "group": `super` consecutive bricks go to the same XCD (compact super-bricks).
      if (interleave && super > 0 && (i + 1) % super == 0) {
        // deal the z-columns of the last `super` bricks round-robin over their workgroups: every workgroup then holds
        // columns from all over the super-brick and meets about the same number of valid voxels in every view
        size_t n = (size_t)super * sx * sy * sz, base = S.vox.size() - n, ncol = n / sz, nw = n / wg, cpw = wg / sz;
        std::vector<int32_t> tmp(S.vox.begin() + base, S.vox.end());
        for (size_t c = 0; c < ncol; ++c) {
          size_t w = c % nw, j = c / nw;
          if (j >= cpw) continue;
          memcpy(&S.vox[base + (w * cpw + j) * sz], &tmp[c * sz], sz * sizeof(int32_t));
        }
      }
      int xcd = sched == "own" ? 0 : (super > 0 ? (int)((i / super) % nx) : (int)(i % nx));
      for (int k = 0; k < per; ++k) S.queue[xcd].push_back((int)(i * per + k));
    }
  }
  int nxs = sched == "own" ? 1 : nx;
  std::vector<Cache> l2;
  for (int x = 0; x < nxs; ++x) l2.emplace_back((size_t)(l2mb * 1024 * 1024), 16);
  Cache mall((size_t)(mallmb * 1024 * 1024), 16);
  // event-driven replay, all XCDs advance on one clock
  struct Ev { double t; int xcd, slot; bool operator<(const Ev& o) const { return t > o.t; } };
  struct Slot { int block = -1, view = 0; };
  std::vector<std::vector<Slot>> slots(nxs, std::vector<Slot>(R));
  std::vector<size_t> qpos(nxs, 0);
  std::priority_queue<Ev> pq;
  for (int x = 0; x < nxs; ++x)
    for (int s = 0; s < R; ++s)
      if (qpos[x] < S.queue[x].size()) { slots[x][s].block = S.queue[x][qpos[x]++]; pq.push({0.0 + 1e-6 * s, x, s}); }
  uint64_t lines = 0;
  std::vector<int> pending;             // lockstep: slots waiting for the whole XCD to finish its blocks
  std::vector<int> done_cnt(nxs, 0), live_cnt(nxs, 0);
  for (int x = 0; x < nxs; ++x) for (auto& s : slots[x]) live_cnt[x] += s.block >= 0;
  std::vector<std::vector<int>> waiting(nxs);
  while (!pq.empty()) {
    Ev e = pq.top(); pq.pop();
    Slot& s = slots[e.xcd][e.slot];
    const int32_t* vx = &S.vox[(size_t)s.block * wg];
    const int32_t* pv = &pix[(size_t)s.view * G];
    int n = 0;
    for (int t = 0; t < wg; ++t) {
      if (vx[t] < 0) continue;
      int p = pv[vx[t]];
      if (p < 0) continue;
      uint32_t line = (uint32_t)s.view * (uint32_t)(H * W) + (uint32_t)p;
      ++n;
      if (!l2[e.xcd].access(line)) mall.access(line);
    }
    lines += n;
    double t1 = e.t + (n ? c0 + n : 1.0);
    if (++s.view < V) { pq.push({t1, e.xcd, e.slot}); continue; }
    s.view = 0;
    if (lockstep) {                       // persistent grid in lockstep: the next round starts when all slots are done
      waiting[e.xcd].push_back(e.slot);
      if ((int)waiting[e.xcd].size() == live_cnt[e.xcd]) {
        int nl = 0;
        for (int sl : waiting[e.xcd]) {
          if (qpos[e.xcd] < S.queue[e.xcd].size()) { slots[e.xcd][sl].block = S.queue[e.xcd][qpos[e.xcd]++]; pq.push({t1, e.xcd, sl}); ++nl; }
          else slots[e.xcd][sl].block = -1;
        }
        live_cnt[e.xcd] = nl;
        waiting[e.xcd].clear();
      }
    } else if (qpos[e.xcd] < S.queue[e.xcd].size()) {
      s.block = S.queue[e.xcd][qpos[e.xcd]++];
      pq.push({t1, e.xcd, e.slot});
    } else s.block = -1;
  }
  uint64_t h = 0, m = 0;
  for (auto& c : l2) { h += c.hit; m += c.miss; }
  double fabric = (double)m * 128 * 8 / 1e9, hbm = (double)mall.miss * 128 * 8 / 1e9;
  printf("%s order=%s brick=%dx%dx%d tt=%d wg=%d R=%d c0=%g lockstep=%d super=%d l2=%gMB: lines/sweep %.1fM  L2 hit %.1f%%  fabric reads %.1f GB"
         "  (+7.25 GB writes = %.1f GB)  MALL hit %.1f%% of L2 misses, HBM reads %.1f GB\n",
         sched.c_str(), border.c_str(), sx, sy, sz, tt, wg, R, c0, lockstep, super, l2mb, lines / 1e6, 100.0 * h / (h + m), fabric,
         fabric + 7.25, 100.0 * mall.hit / std::max<uint64_t>(1, mall.hit + mall.miss), hbm);
  return 0;
}
