// Exercises include/plvs_hip.hpp (the C++ mirror of the PLVS interfaces) end to end, prints one line per
// result (sizes + FNV-1a hashes) and dumps the raw arrays into <out_dir>; tests/test_cpp_mirror.py compares
// them with the same calls made through the Python mirror (which the parity tests pin against the oracle).
// Usage: mirror_smoke <left.pgm> <right.pgm> <out_dir>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>

#include "plvs_hip.hpp"

using namespace PLVS2hip;

static uint64_t fnv(const void* p, size_t n, uint64_t h = 1469598103934665603ull) {
  const uint8_t* b = static_cast<const uint8_t*>(p);
  for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}

static std::string g_out;
static void dump(const char* name, const void* p, size_t n) {
  std::ofstream f(g_out + "/" + name + ".bin", std::ios::binary);
  f.write(static_cast<const char*>(p), (std::streamsize)n);
}

static std::vector<uint8_t> read_pgm(const char* path, int* w, int* h) {
  std::ifstream f(path, std::ios::binary);
  if (!f) { std::fprintf(stderr, "cannot open %s\n", path); std::exit(2); }
  std::string magic;
  int maxv;
  f >> magic >> *w >> *h >> maxv;
  f.get();
  std::vector<uint8_t> img((size_t)*w * *h);
  f.read(reinterpret_cast<char*>(img.data()), (std::streamsize)img.size());
  return img;
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  g_out = argv[3];
  int w, h, w2, h2;
  std::vector<uint8_t> left = read_pgm(argv[1], &w, &h), right = read_pgm(argv[2], &w2, &h2);
  Image8U il{h, w, (size_t)w, left.data()}, ir{h2, w2, (size_t)w2, right.data()};

  // ---- ORB on both images, stereo matches
  ORBextractor exl(2000, 1.2f, 8, 20, 7), exr(2000, 1.2f, 8, 20, 7);
  std::vector<KeyPoint> kl, kr;
  std::vector<uint8_t> dl, dr;
  const int mono = exl(il, kl, dl);
  exr(ir, kr, dr);
  std::printf("orb_left %d %d %016llx %016llx\n", mono, (int)kl.size(), (unsigned long long)fnv(kl.data(), kl.size() * sizeof(KeyPoint)),
              (unsigned long long)fnv(dl.data(), dl.size()));
  dump("orb_keys", kl.data(), kl.size() * sizeof(KeyPoint));
  dump("orb_desc", dl.data(), dl.size());
  std::vector<float> uR, depth;
  ComputeStereoMatches(exl, exr, kl, dl, kr, dr, 386.1448f / 718.856f, 386.1448f, uR, depth);
  std::printf("stereo %016llx %016llx\n", (unsigned long long)fnv(uR.data(), uR.size() * 4), (unsigned long long)fnv(depth.data(), depth.size() * 4));
  dump("stereo_uright", uR.data(), uR.size() * 4);
  dump("stereo_depth", depth.data(), depth.size() * 4);
  Image8U none;
  std::vector<KeyPoint> ke;
  std::vector<uint8_t> de;
  std::printf("orb_empty %d %d\n", exl(none, ke, de), (int)ke.size());

  // ---- lines on the left image, LBD k-NN left vs right lines
  LineExtractor lx(100), lx2(100);
  std::vector<KeyLine> ll, lr;
  std::vector<uint8_t> ldl, ldr;
  lx(il, ll, ldl);
  lx2(ir, lr, ldr);
  std::printf("lines %d %016llx %016llx\n", (int)ll.size(), (unsigned long long)fnv(ll.data(), ll.size() * sizeof(KeyLine)),
              (unsigned long long)fnv(ldl.data(), ldl.size()));
  dump("lines_keys", ll.data(), ll.size() * sizeof(KeyLine));
  dump("lines_desc", ldl.data(), ldl.size());
  {   // the same extractor with Line.LSD.on: 1, options as Tracking fills them
    LineExtractor::UseLsdExtractor() = true;
    LSDOptions lo;
    lo.refine = 1; lo.log_eps = 1.0; lo.density_th = 0.6; lo.min_length = 0.025;
    LineExtractor lsd(100, lo);
    LineExtractor::UseLsdExtractor() = false;
    std::vector<KeyLine> kl;
    std::vector<uint8_t> kd;
    lsd(il, kl, kd);
    std::printf("lsd_lines %d\n", (int)kl.size());
    dump("lsd_keys", kl.data(), kl.size() * sizeof(KeyLine));
    dump("lsd_desc", kd.data(), kd.size());
  }
  BinaryDescriptorMatcher bdm;
  std::vector<std::vector<DMatch>> matches;
  bdm.knnMatch(ldl.data(), (int)ll.size(), ldr.data(), (int)lr.size(), matches);
  uint64_t hm = 1469598103934665603ull;
  std::vector<int32_t> flat;
  for (const auto& row : matches)
    for (const DMatch& m : row) {
      hm = fnv(&m.queryIdx, 4, hm); hm = fnv(&m.trainIdx, 4, hm); hm = fnv(&m.distance, 4, hm);
      flat.push_back(m.queryIdx); flat.push_back(m.trainIdx); flat.push_back((int32_t)m.distance);
    }
  dump("knn", flat.data(), flat.size() * 4);
  std::printf("knn %d %016llx %d\n", (int)matches.size(), (unsigned long long)hm,
              ORBmatcher::DescriptorDistance(dl.data(), dl.data() + 32));

  // ---- the search functions: lines left (query) vs right (train); ORB BoW search with the vocabulary node = first
  // descriptor byte (a stand-in for DBoW2's FeatureVector)
  {
    std::vector<uint8_t> valid(ll.size(), 1);
    std::vector<float> al(ll.size()), ar(lr.size());
    std::vector<int32_t> ol(ll.size()), orr(lr.size());
    for (size_t i = 0; i < ll.size(); ++i) { al[i] = ll[i].angle; ol[i] = ll[i].octave; if (i % 7 == 3) valid[i] = 0; }
    for (size_t i = 0; i < lr.size(); ++i) { ar[i] = lr[i].angle; orr[i] = lr[i].octave; }
    LineMatcher lm(0.8f, true);
    std::vector<int32_t> a1, a2;
    const int n1 = lm.SearchByKnn(ldl.data(), (int)ll.size(), valid.data(), al.data(), ldr.data(), (int)lr.size(), ar.data(), a1);
    const int n2 = lm.SearchByKnnKF(ldl.data(), (int)ll.size(), valid.data(), al.data(), ldr.data(), (int)lr.size(), ar.data(), a2);
    std::vector<DMatch> vm;
    std::vector<bool> vv;
    const int n3 = lm.SearchStereoMatchesByKnn(ldl.data(), (int)ll.size(), al.data(), ol.data(), ldr.data(), (int)lr.size(),
                                               ar.data(), orr.data(), vm, vv);
    std::vector<int32_t> sflat;
    for (size_t i = 0; i < vm.size(); ++i) { sflat.push_back(vm[i].queryIdx); sflat.push_back(vm[i].trainIdx); sflat.push_back((int32_t)vm[i].distance); sflat.push_back(vv[i] ? 1 : 0); }
    std::printf("line_search %d %d %d %d\n", n1, n2, n3, (int)vm.size());
    dump("line_ff", a1.data(), a1.size() * 4);
    dump("line_kf", a2.data(), a2.size() * 4);
    dump("line_stereo", sflat.data(), sflat.size() * 4);

    auto featvec = [](const std::vector<uint8_t>& desc, std::vector<uint32_t>& ids, std::vector<int32_t>& off, std::vector<uint32_t>& idx) {
      std::map<uint32_t, std::vector<uint32_t>> fv;
      for (size_t i = 0; i < desc.size() / 32; ++i) fv[desc[32 * i] >> 2].push_back((uint32_t)i);
      off.push_back(0);
      for (const auto& kv : fv) { ids.push_back(kv.first); idx.insert(idx.end(), kv.second.begin(), kv.second.end()); off.push_back((int32_t)idx.size()); }
    };
    std::vector<uint32_t> kid, kidx, fid, fidx;
    std::vector<int32_t> koff, foff;
    featvec(dl, kid, koff, kidx);
    featvec(dr, fid, foff, fidx);
    plvs_featvec_view kv{(int32_t)kid.size(), kid.data(), koff.data(), kidx.data()}, fv{(int32_t)fid.size(), fid.data(), foff.data(), fidx.data()};
    std::vector<uint8_t> kvalid(kl.size(), 1);
    std::vector<float> ka(kl.size()), fa(kr.size());
    for (size_t i = 0; i < kl.size(); ++i) ka[i] = kl[i].angle;
    for (size_t i = 0; i < kr.size(); ++i) fa[i] = kr[i].angle;
    std::vector<int32_t> ab;
    const int nb = ORBmatcher(0.7f, true).SearchByBoW(kv, dl.data(), (int)kl.size(), kvalid.data(), ka.data(), fv, dr.data(),
                                                     (int)kr.size(), fa.data(), ab);
    std::printf("bow %d\n", nb);
    dump("bow", ab.data(), ab.size() * 4);
  }

  // ---- dense stereo on the same pair (width cropped to 1240)
  {
    const int sw = 1240;
    std::vector<uint8_t> cl((size_t)sw * h), cr((size_t)sw * h), disp((size_t)sw * h);
    for (int y = 0; y < h; ++y) { std::memcpy(&cl[(size_t)y * sw], &left[(size_t)y * w], sw); std::memcpy(&cr[(size_t)y * sw], &right[(size_t)y * w], sw); }
    StereoSGM sgm(sw, h, 64);
    sgm.execute(cl.data(), cr.data(), disp.data());
    size_t valid = 0;
    for (uint8_t d : disp) valid += d > 0;
    std::printf("sgm %d %016llx\n", (int)valid, (unsigned long long)fnv(disp.data(), disp.size()));
    dump("sgm", disp.data(), disp.size());
  }

  // ---- depth -> cloud -> chisel map (with carving) -> meshes -> output cloud; voxblox beside it
  const int W = 320, H = 240;
  const double fx = 258.65, fy = 258.23, cx = 159.3, cy = 127.6;
  std::vector<float> dimg((size_t)W * H);
  std::vector<uint8_t> cimg((size_t)W * H * 3);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      dimg[(size_t)y * W + x] = 1.5f + 0.002f * (float)x + 0.3f * std::sin(0.05f * (float)y);
      for (int c = 0; c < 3; ++c) cimg[((size_t)y * W + x) * 3 + c] = (uint8_t)((x * 3 + y * 5 + c * 40) & 255);
    }
  for (int y = 40; y < 60; ++y)
    for (int x = 100; x < 140; ++x) dimg[(size_t)y * W + x] = 0.0f;   // a hole
  PointCloudGenerator gen(W, H, 2, fx, fy, cx, cy, 0.1, 5.0);
  Image8U ci{H, W, (size_t)W * 3, cimg.data()};
  Image32F di{H, W, (size_t)W * sizeof(float), dimg.data()};
  std::vector<int32_t> p2p;
  std::vector<PointSurfelSegment> cloud = gen.GeneratePointCloudInCameraFrameBGRA(7, ci, di, &p2p);
  std::printf("cloud %d %016llx %016llx\n", (int)cloud.size(), (unsigned long long)fnv(cloud.data(), cloud.size() * sizeof(PointSurfelSegment)),
              (unsigned long long)fnv(p2p.data(), p2p.size() * 4));
  dump("cloud", cloud.data(), cloud.size() * sizeof(PointSurfelSegment));
  dump("p2p", p2p.data(), p2p.size() * 4);
  dump("depth_img", dimg.data(), dimg.size() * 4);
  dump("color_img", cimg.data(), cimg.size());
  SE3f Twc = {{1, 0, 0, 0.1f, 0, 1, 0, -0.2f, 0, 0, 1, 0.05f}};
  PointCloudMapChisel map(0.05f, true);
  map.InsertCloudWithDepth(cloud, Twc, di, (float)fx, (float)fy, (float)cx, (float)cy);
  Twc.m[3] += 0.03f;
  map.InsertCloudWithDepth(cloud, Twc, di, (float)fx, (float)fy, (float)cx, (float)cy);
  const int npts = map.UpdateMap();
  uint64_t hv = 1469598103934665603ull;
  for (const auto& kv : map.GetAllMeshes()) {
    hv = fnv(kv.second.vertices.data(), kv.second.vertices.size() * 4, hv);
    hv = fnv(kv.second.normals.data(), kv.second.normals.size() * 4, hv);
    hv = fnv(kv.second.colors.data(), kv.second.colors.size() * 4, hv);
    hv = fnv(kv.second.kfids.data(), kv.second.kfids.size() * 4, hv);
  }
  dump("map_cloud", map.GetPointCloud().data(), map.GetPointCloud().size() * sizeof(PointSurfelSegment));
  std::printf("chisel %d %d %016llx\n", npts, (int)map.GetAllMeshes().size(), (unsigned long long)hv);
  PointCloudMapVoxblox vmap(0.05f, false, "simple");
  vmap.InsertCloud(cloud, Twc);
  Twc.m[3] -= 0.03f;
  vmap.InsertCloud(cloud, Twc);
  const int vpts = vmap.UpdateMap();
  dump("vmap_cloud", vmap.GetPointCloud().data(), vmap.GetPointCloud().size() * sizeof(PointSurfelSegment));
  std::printf("voxblox %d %d %d\n", vmap.NumBlocks(), vpts, (int)vmap.GetMeshLayer().size());
  {   // PLVS's YAML default method
    PointCloudMapVoxblox vfast(0.05f);   // (the constructor's default, as the reference's: src/PointCloudMapVoxblox.cc:44)
    SE3f Tf = Twc;
    vfast.InsertCloud(cloud, Tf);
    Tf.m[3] += 0.03f;
    vfast.InsertCloud(cloud, Tf);
    const int fpts = vfast.UpdateMap();
    dump("vfast_cloud", vfast.GetPointCloud().data(), vfast.GetPointCloud().size() * sizeof(PointSurfelSegment));
    std::printf("voxblox_fast %d %d\n", vfast.NumBlocks(), fpts);
  }
  {   // saveMap / loadMap of the layer: into an empty map, meshed there
    PointCloudMapVoxblox vcopy(0.05f, false, "simple");
    vcopy.LoadLayer(vmap.SaveLayer());
    const int cpts = vcopy.UpdateMap();
    std::printf("voxblox_layer %d %d %d\n", vcopy.NumBlocks(), cpts,
                (int)(cpts == vpts && std::memcmp(vcopy.GetPointCloud().data(), vmap.GetPointCloud().data(),
                                                  (size_t)vpts * sizeof(PointSurfelSegment)) == 0));
  }
  {   // LoadMap: the chisel map's own output cloud back into an empty map, along its normals
    PointCloudMapChisel again(0.05f);
    const int lpts = again.LoadMap(map.GetPointCloud());
    dump("loaded_cloud", again.GetPointCloud().data(), again.GetPointCloud().size() * sizeof(PointSurfelSegment));
    std::printf("loadmap %d %d\n", lpts, (int)again.GetAllMeshes().size());
  }
  {   // OnMapChange with cloud deformation: the volume and the stored meshes move with their key frame (kfid 7 here)
    PointCloudMapChisel dmap(0.05f, false, 0.05f, 0.05f, 5.0f, /*reset*/ false, /*deform*/ true);
    SE3f T = {{1, 0, 0, 0.1f, 0, 1, 0, -0.2f, 0, 0, 1, 0.05f}};
    dmap.InsertCloud(cloud, T);
    dmap.UpdateMap();
    std::map<uint32_t, PointCloudMapChisel::Rt12> rt;
    rt[7] = PointCloudMapChisel::Rt12{{0.9998f, -0.02f, 0.f, 0.02f, 0.9998f, 0.f, 0.f, 0.f, 1.f, 0.04f, -0.03f, 0.11f}};
    const int dpts = dmap.OnMapChange(rt);
    dump("deformed_cloud", dmap.GetPointCloud().data(), dmap.GetPointCloud().size() * sizeof(PointSurfelSegment));
    dmap.InsertCloud(cloud, T);   // the deformed volume takes the next cloud
    const int dpts2 = dmap.UpdateMap();
    dump("deformed_cloud2", dmap.GetPointCloud().data(), dmap.GetPointCloud().size() * sizeof(PointSurfelSegment));
    std::printf("deform %d %d\n", dpts, dpts2);
  }
  map.Clear();
  std::printf("cleared %d\n", map.UpdateMap());
  return 0;
}
