// Host-only tests of the facade's point-cloud file I/O (include/sycl_points/amd/io.hpp), modelled on the reference's
// cpp/tests/test_file_io.cpp: write -> read round trips in the four format x encoding combinations, the rgb reading
// known answers (PLY uchar red/green/blue, PCD one-byte r/g/b -> value / 255), cross-format conversion, and the error
// cases. No device work: runs in the CPU test suite (tests/test_io_cpu.py). Exit code 0 = all checks passed.
#include <cstdio>
#include <fstream>
#include <random>

#include "sycl_points/io/point_cloud_reader.hpp"
#include "sycl_points/io/point_cloud_writer.hpp"

using namespace sycl_points;

static int g_failed = 0, g_checks = 0;
#define CHECK(cond)                                                                                          \
    do {                                                                                                     \
        ++g_checks;                                                                                          \
        if (!(cond)) { ++g_failed; std::printf("  CHECK FAILED %s:%d  %s\n", __FILE__, __LINE__, #cond); }   \
    } while (0)

static const std::string kDir = "/tmp/sp_io_test_";

// test_file_io.cpp:24-45 generateTestData: uniform points in [-10, 10]^3, mt19937(42)
static PointCloudCPU make_cloud(size_t n, bool rgb = false, bool intensity = false) {
    PointCloudCPU c;
    std::mt19937 gen(42);
    std::uniform_real_distribution<float> U(-10.0f, 10.0f), C01(0.0f, 1.0f);
    for (size_t i = 0; i < n; ++i) c.points->emplace_back(U(gen), U(gen), U(gen), 1.0f);
    if (rgb)
        for (size_t i = 0; i < n; ++i) c.rgb->emplace_back(C01(gen), C01(gen), C01(gen), 1.0f);
    if (intensity)
        for (size_t i = 0; i < n; ++i) c.intensities->push_back(U(gen));
    return c;
}

// test_file_io.cpp:47-63 comparePointClouds (tolerance 1e-5 for ascii: six decimals; exact for binary)
static bool same_points(const PointCloudCPU& a, const PointCloudCPU& b, float tol) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); ++i)
        for (int k = 0; k < 4; ++k)
            if (!(std::fabs((*a.points)[i][k] - (*b.points)[i][k]) <= tol)) return false;
    return true;
}

static void round_trips() {  // test_file_io.cpp:88-105, 226-251
    const PointCloudCPU cloud = make_cloud(1000);
    for (const char* ext : {"ply", "pcd"})
        for (const bool binary : {false, true}) {
            const std::string f = kDir + (binary ? "bin." : "ascii.") + ext;
            PointCloudWriter::writeFile(f, cloud, binary);
            const PointCloudCPU back = PointCloudReader::readFile(f);
            CHECK(same_points(cloud, back, binary ? 0.0f : 1e-5f));
            CHECK(!back.has_rgb() && !back.has_intensity());
            std::remove(f.c_str());
        }
    // writePLY / writePCD append the extension when it is missing (point_cloud_writer.hpp:370-398)
    PointCloudWriter::writePLY(kDir + "noext", cloud, true);
    CHECK(same_points(cloud, PointCloudReader::readFile(kDir + "noext.ply"), 0.0f));
    PointCloudWriter::writePCD(kDir + "noext", cloud, true);
    CHECK(same_points(cloud, PointCloudReader::readFile(kDir + "noext.pcd"), 0.0f));
    std::remove((kDir + "noext.ply").c_str());
    std::remove((kDir + "noext.pcd").c_str());
}

static void attributes() {
    // PLY carries rgb (uchar, value = clamp(c) * 255 truncated) and intensity (float); reading divides by 255
    const PointCloudCPU cloud = make_cloud(500, true, true);
    for (const bool binary : {false, true}) {
        const std::string f = kDir + "attr.ply";
        PointCloudWriter::writeFile(f, cloud, binary);
        const PointCloudCPU back = PointCloudReader::readFile(f);
        CHECK(back.has_rgb() && back.has_intensity() && same_points(cloud, back, binary ? 0.0f : 1e-5f));
        bool ok = back.has_rgb() && back.has_intensity();
        for (size_t i = 0; ok && i < cloud.size(); ++i) {
            for (int k = 0; k < 3; ++k) {
                const float expect = (float)(uint8_t)((*cloud.rgb)[i][k] * 255.f) / 255.f;
                ok = ok && (*back.rgb)[i][k] == expect;
            }
            ok = ok && (*back.rgb)[i][3] == 1.0f;
            ok = ok && std::fabs((*back.intensities)[i] - (*cloud.intensities)[i]) <= (binary ? 0.0f : 1e-5f);
        }
        CHECK(ok);
        // read_rgb / read_intensity off: attributes are skipped, points unchanged
        const PointCloudCPU bare = PointCloudReader::readFile(f, false, false);
        CHECK(!bare.has_rgb() && !bare.has_intensity() && same_points(cloud, bare, binary ? 0.0f : 1e-5f));
        std::remove(f.c_str());
    }
    // PCD writes ONE packed `rgb` word (0x00RRGGBB) and no intensity; the reader only decodes one-byte r/g/b fields, so a
    // written PCD reads back without colour (both as in the reference)
    for (const bool binary : {false, true}) {
        const std::string f = kDir + "attr.pcd";
        PointCloudWriter::writeFile(f, cloud, binary);
        std::ifstream in(f, std::ios::binary);
        std::string head((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        CHECK(head.find("FIELDS x y z rgb\n") != std::string::npos && head.find("TYPE F F F U\n") != std::string::npos);
        const PointCloudCPU back = PointCloudReader::readFile(f);
        CHECK(!back.has_rgb() && same_points(cloud, back, binary ? 0.0f : 1e-5f));
        std::remove(f.c_str());
    }
}

static void rgb_known_answers() {  // test_file_io.cpp:253-392: (255,0,0) and (0,255,128) -> (1,0,0) and (0,1,128/255)
    const float xyz[2][3] = {{0, 0, 0}, {1, 2, 3}};
    const uint8_t col[2][3] = {{255, 0, 0}, {0, 255, 128}};
    auto check = [&](const std::string& f) {
        const PointCloudCPU c = PointCloudReader::readFile(f);
        CHECK(c.has_rgb() && c.size() == 2);
        if (c.has_rgb() && c.size() == 2) {
            CHECK((*c.rgb)[0].x() == 1.0f && (*c.rgb)[0].y() == 0.0f && (*c.rgb)[0].z() == 0.0f);
            CHECK((*c.rgb)[1].x() == 0.0f && (*c.rgb)[1].y() == 1.0f && std::fabs((*c.rgb)[1].z() - 128.f / 255.f) < 1e-6f);
            CHECK((*c.points)[1].x() == 1.0f && (*c.points)[1].y() == 2.0f && (*c.points)[1].z() == 3.0f);
        }
        std::remove(f.c_str());
    };
    const std::string ply_head = "element vertex 2\nproperty float x\nproperty float y\nproperty float z\n"
                                 "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n";
    const std::string pcd_head = "# .PCD v0.7 - Point Cloud Data file format\nFIELDS x y z r g b\nSIZE 4 4 4 1 1 1\n"
                                 "TYPE F F F U U U\nCOUNT 1 1 1 1 1 1\nWIDTH 2\nHEIGHT 1\nPOINTS 2\n";
    for (const bool ply : {true, false})
        for (const bool binary : {false, true}) {
            const std::string f = kDir + std::string("rgb.") + (ply ? "ply" : "pcd");
            std::ofstream o(f, std::ios::binary);
            if (ply) o << "ply\nformat " << (binary ? "binary_little_endian" : "ascii") << " 1.0\n" << ply_head;
            else o << pcd_head << "DATA " << (binary ? "binary" : "ascii") << "\n";
            for (int i = 0; i < 2; ++i) {
                if (binary) {
                    o.write(reinterpret_cast<const char*>(xyz[i]), 12);
                    o.write(reinterpret_cast<const char*>(col[i]), 3);
                } else {
                    o << xyz[i][0] << " " << xyz[i][1] << " " << xyz[i][2] << " " << (int)col[i][0] << " " << (int)col[i][1]
                      << " " << (int)col[i][2] << "\n";
                }
            }
            o.close();
            check(f);
        }
}

static void edge_cases() {  // test_file_io.cpp:394-470
    bool threw = false;
    try { PointCloudWriter::writeFile(kDir + "empty.ply", PointCloudCPU(), false); } catch (const std::runtime_error&) { threw = true; }
    CHECK(threw);  // EmptyPointCloud
    PointCloudCPU one;
    one.points->emplace_back(1.23f, 4.56f, 7.89f, 1.0f);
    for (const char* ext : {"ply", "pcd"}) {  // SinglePoint
        PointCloudWriter::writeFile(kDir + "single." + ext, one, false);
        CHECK(same_points(one, PointCloudReader::readFile(kDir + "single." + ext), 1e-5f));
        std::remove((kDir + "single." + ext).c_str());
    }
    const PointCloudCPU big = make_cloud(10000);  // LargePointCloud
    PointCloudWriter::writeFile(kDir + "large.ply", big, true);
    CHECK(same_points(big, PointCloudReader::readFile(kDir + "large.ply"), 0.0f));
    std::remove((kDir + "large.ply").c_str());
    threw = false;  // InvalidFileFormat
    try { PointCloudWriter::writeFile(kDir + "bad.xyz", big, false); } catch (const std::runtime_error&) { threw = true; }
    CHECK(threw);
    std::remove((kDir + "bad.xyz").c_str());
    threw = false;
    try { PointCloudReader::readFile(kDir + "bad.xyz"); } catch (const std::runtime_error&) { threw = true; }
    CHECK(threw);
    threw = false;  // unwritable path
    try { PointCloudWriter::writeFile("/nonexistent_dir_sp/x.ply", big, false); } catch (const std::runtime_error&) { threw = true; }
    CHECK(threw);
    // points with non-finite coordinates are skipped, and a cloud of only such points is refused
    PointCloudCPU holes = make_cloud(10);
    (*holes.points)[3].x() = std::numeric_limits<float>::quiet_NaN();
    (*holes.points)[7].z() = std::numeric_limits<float>::infinity();
    PointCloudWriter::writeFile(kDir + "holes.pcd", holes, true);
    const PointCloudCPU back = PointCloudReader::readFile(kDir + "holes.pcd");
    CHECK(back.size() == 8 && (*back.points)[3].x() == (*holes.points)[4].x());
    std::remove((kDir + "holes.pcd").c_str());
    PointCloudCPU none;
    none.points->emplace_back(std::numeric_limits<float>::quiet_NaN(), 0.f, 0.f, 1.f);
    threw = false;
    try { PointCloudWriter::writeFile(kDir + "none.ply", none, false); } catch (const std::runtime_error&) { threw = true; }
    CHECK(threw);
    std::remove((kDir + "none.ply").c_str());
}

static void bundled_cloud_cross_format(const char* dir) {  // test_file_io.cpp:155-222 on cpp/data/source.ply
    const PointCloudCPU src = PointCloudReader::readFile(std::string(dir) + "/source.ply");
    CHECK(src.size() == 69792 && src.has_intensity());
    for (const char* ext : {"pcd", "ply"})
        for (const bool binary : {false, true}) {
            const std::string f = kDir + std::string("cross.") + ext;
            PointCloudWriter::writeFile(f, src, binary);
            CHECK(same_points(src, PointCloudReader::readFile(f), binary ? 0.0f : 1e-5f * 100.f));  // |coords| < 100: 6 decimals
            std::remove(f.c_str());
        }
}

int main() {
    round_trips();
    attributes();
    rgb_known_answers();
    edge_cases();
    if (const char* dir = std::getenv("SP_GOLDEN_DIR")) bundled_cloud_cross_format(dir);
    std::printf("test_io: %d checks, %d failed\n", g_checks, g_failed);
    return g_failed ? 1 : 0;
}
