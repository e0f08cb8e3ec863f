// sycl_points facade for MI355X — point-cloud file I/O either side of the hot path:
//   PointCloudReader (io/point_cloud_reader.hpp:20-548): PLY `element vertex` with scalar properties (ascii or
//     binary_little_endian) and PCD (`DATA ascii` / `DATA binary`): x, y, z (any scalar type) into PointCloudCPU::points
//     (w = 1); `intensity` / `scalar_intensity` into intensities; red/green/blue (PLY) or one-byte r/g/b fields (PCD) into
//     rgb as value / 255 with alpha 1 — each when asked for (both default on, as in the reference).
//   PointCloudWriter (io/point_cloud_writer.hpp:14-400): PLY and PCD, ascii (fixed, 6 decimals) or binary, points with
//     non-finite coordinates skipped, rgb as uchar triples (PLY) or one packed `rgb` word (PCD), intensity (PLY only).
// Pure host code: a PointCloudShared is read through its host view (one device->host copy if the device side is newer).
#pragma once
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <fstream>
#include <iomanip>
#include <sstream>

#include "core.hpp"

namespace sycl_points {

class PointCloudReader {
public:
    /// io/point_cloud_reader.hpp:494-526
    static PointCloudCPU readFile(const std::string& filename, bool read_rgb = true, bool read_intensity = true) {
        const auto dot = filename.find_last_of('.');
        std::string ext = dot == std::string::npos ? "" : filename.substr(dot + 1);
        for (auto& c : ext) c = (char)std::tolower((unsigned char)c);
        if (ext == "ply") return readPLY(filename, read_intensity, read_rgb);
        if (ext == "pcd") return readPCD(filename, read_intensity, read_rgb);
        throw std::runtime_error("[PointCloudReader::readFile] Unsupported file format: " + filename);
    }
    /// io/point_cloud_reader.hpp:528-548: straight into a PointCloudShared of `queue`.
    static PointCloudShared readFile(const std::string& filename, const sycl_utils::DeviceQueue& queue, bool read_rgb = true,
                                     bool read_intensity = true) {
        return PointCloudShared(queue, readFile(filename, read_rgb, read_intensity));
    }
    static PointCloudShared readFile(const sycl_utils::DeviceQueue& queue, const std::string& filename, bool read_rgb = true,
                                     bool read_intensity = true) {
        return readFile(filename, queue, read_rgb, read_intensity);
    }

    /// PCD (io/point_cloud_reader.hpp:278-492): header FIELDS / SIZE / TYPE / COUNT / WIDTH / HEIGHT / POINTS / DATA,
    /// `DATA ascii` or `DATA binary` (little endian rows of the declared fields). x, y, z are required; `intensity`
    /// is read when asked for. `binary_compressed` is rejected (as in the reference, which only knows ascii and binary).
    static PointCloudCPU readPCD(const std::string& filename, bool read_intensity = true, bool read_rgb = true) {
        std::ifstream f(filename, std::ios::binary);
        if (!f) throw std::runtime_error("[PointCloudReader::readPCD] cannot open " + filename);
        struct Field { std::string name; size_t size = 4; char type = 'F'; size_t count = 1; };
        std::vector<Field> fields;
        size_t n_points = 0, width = 0, height = 1;
        std::string line, data;
        bool have_points = false;
        while (std::getline(f, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (line.empty() || line[0] == '#') continue;
            std::istringstream ss(line);
            std::string key;
            ss >> key;
            if (key == "FIELDS") {
                std::string name;
                while (ss >> name) { Field fd; fd.name = name; fields.push_back(fd); }
            } else if (key == "SIZE") {
                for (auto& fd : fields) ss >> fd.size;
            } else if (key == "TYPE") {
                for (auto& fd : fields) ss >> fd.type;
            } else if (key == "COUNT") {
                for (auto& fd : fields) ss >> fd.count;
            } else if (key == "WIDTH") {
                ss >> width;
            } else if (key == "HEIGHT") {
                ss >> height;
            } else if (key == "POINTS") {
                ss >> n_points;
                have_points = true;
            } else if (key == "DATA") {
                ss >> data;
                break;  // DATA is the last header line
            }
        }
        if (!have_points) n_points = width * height;
        if (fields.empty() || data.empty()) throw std::runtime_error("[PointCloudReader::readPCD] Invalid PCD format: missing point data");
        if (data != "ascii" && data != "binary")
            throw std::runtime_error("[PointCloudReader::readPCD] unsupported DATA '" + data + "'");
        int ix = -1, iy = -1, iz = -1, ii = -1, ir = -1, ig = -1, ib = -1;
        std::vector<size_t> first(fields.size());  // index of a field's first value in a row's value list
        size_t n_vals = 0, stride = 0;
        for (size_t i = 0; i < fields.size(); ++i) {
            first[i] = n_vals;
            n_vals += fields[i].count;
            stride += fields[i].size * fields[i].count;
            if (fields[i].name == "x") ix = (int)first[i];
            else if (fields[i].name == "y") iy = (int)first[i];
            else if (fields[i].name == "z") iz = (int)first[i];
            else if (fields[i].name == "intensity") ii = (int)first[i];
            // colour: three one-byte fields r, g, b (io/point_cloud_reader.hpp:309-351; a packed `rgb` word is not decoded)
            else if (fields[i].name == "r" && fields[i].size == 1) ir = (int)first[i];
            else if (fields[i].name == "g" && fields[i].size == 1) ig = (int)first[i];
            else if (fields[i].name == "b" && fields[i].size == 1) ib = (int)first[i];
        }
        if (ix < 0 || iy < 0 || iz < 0) throw std::runtime_error("[PointCloudReader::readPCD] x/y/z fields missing");
        PointCloudCPU cloud;
        cloud.points->resize(n_points);
        const bool want_i = read_intensity && ii >= 0;
        if (want_i) cloud.intensities->resize(n_points);
        const bool want_c = read_rgb && ir >= 0 && ig >= 0 && ib >= 0;
        if (want_c) cloud.rgb->resize(n_points);
        const int rgb3[3] = {want_c ? ir : -1, ig, ib};
        std::vector<double> vals(n_vals);
        if (data == "ascii") {
            for (size_t v = 0; v < n_points; ++v) {
                for (size_t k = 0; k < n_vals; ++k)
                    if (!(f >> vals[k])) throw std::runtime_error("[PointCloudReader::readPCD] Error reading ascii PCD data");
                store(cloud, v, vals, ix, iy, iz, want_i ? ii : -1, rgb3);
            }
        } else {
            std::vector<char> row(stride);
            for (size_t v = 0; v < n_points; ++v) {
                f.read(row.data(), (std::streamsize)stride);
                if (!f) throw std::runtime_error("[PointCloudReader::readPCD] Error reading binary PCD data");
                size_t off = 0, k = 0;
                for (const auto& fd : fields)
                    for (size_t c = 0; c < fd.count; ++c) {
                        vals[k++] = decode_pcd(row.data() + off, fd.type, fd.size);
                        off += fd.size;
                    }
                store(cloud, v, vals, ix, iy, iz, want_i ? ii : -1, rgb3);
            }
        }
        return cloud;
    }

    static PointCloudCPU readPLY(const std::string& filename, bool read_intensity = true, bool read_rgb = true) {
        std::ifstream f(filename, std::ios::binary);
        if (!f) throw std::runtime_error("[PointCloudReader::readPLY] cannot open " + filename);
        struct Prop { std::string type, name; size_t bytes; };
        std::vector<Prop> props;
        std::string line, format;
        size_t n_vertex = 0;
        bool in_vertex = false;
        if (!std::getline(f, line) || line.substr(0, 3) != "ply") throw std::runtime_error("[readPLY] not a PLY file: " + filename);
        while (std::getline(f, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            std::istringstream ss(line);
            std::string tok;
            ss >> tok;
            if (tok == "format") ss >> format;
            else if (tok == "element") {
                std::string name; size_t cnt;
                ss >> name >> cnt;
                in_vertex = (name == "vertex");
                if (in_vertex) n_vertex = cnt;
            } else if (tok == "property" && in_vertex) {
                Prop p;
                ss >> p.type >> p.name;
                if (p.type == "list") throw std::runtime_error("[readPLY] list properties on vertices are not supported");
                p.bytes = type_size(p.type);
                props.push_back(p);
            } else if (tok == "end_header") break;
        }
        if (format != "ascii" && format != "binary_little_endian")
            throw std::runtime_error("[readPLY] unsupported PLY format '" + format + "'");
        int ix = -1, iy = -1, iz = -1, ii = -1, ir = -1, ig = -1, ib = -1;
        for (size_t i = 0; i < props.size(); ++i) {
            if (props[i].name == "x") ix = (int)i;
            else if (props[i].name == "y") iy = (int)i;
            else if (props[i].name == "z") iz = (int)i;
            else if (props[i].name == "intensity" || props[i].name == "scalar_intensity") ii = (int)i;
            else if (props[i].name == "red") ir = (int)i;  // io/point_cloud_reader.hpp:64-66, 197-200
            else if (props[i].name == "green") ig = (int)i;
            else if (props[i].name == "blue") ib = (int)i;
        }
        if (ix < 0 || iy < 0 || iz < 0) throw std::runtime_error("[readPLY] x/y/z properties missing");
        PointCloudCPU cloud;
        cloud.points->resize(n_vertex);
        const bool want_i = read_intensity && ii >= 0;
        if (want_i) cloud.intensities->resize(n_vertex);
        const bool want_c = read_rgb && ir >= 0 && ig >= 0 && ib >= 0;
        if (want_c) cloud.rgb->resize(n_vertex);
        const int rgb3[3] = {want_c ? ir : -1, ig, ib};
        std::vector<double> vals(props.size());
        if (format == "ascii") {
            for (size_t v = 0; v < n_vertex; ++v) {
                for (size_t p = 0; p < props.size(); ++p) f >> vals[p];
                store(cloud, v, vals, ix, iy, iz, want_i ? ii : -1, rgb3);
            }
        } else {
            size_t stride = 0;
            for (auto& p : props) stride += p.bytes;
            std::vector<char> row(stride);
            for (size_t v = 0; v < n_vertex; ++v) {
                f.read(row.data(), (std::streamsize)stride);
                if (!f) throw std::runtime_error("[readPLY] truncated file " + filename);
                size_t off = 0;
                for (size_t p = 0; p < props.size(); ++p) { vals[p] = decode(row.data() + off, props[p].type); off += props[p].bytes; }
                store(cloud, v, vals, ix, iy, iz, want_i ? ii : -1, rgb3);
            }
        }
        return cloud;
    }

private:
    static size_t type_size(const std::string& t) {
        if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
        if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
        if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
        if (t == "double" || t == "float64") return 8;
        throw std::runtime_error("[readPLY] unknown property type " + t);
    }
    static double decode(const char* p, const std::string& t) {
        if (t == "float" || t == "float32") { float v; std::memcpy(&v, p, 4); return v; }
        if (t == "double" || t == "float64") { double v; std::memcpy(&v, p, 8); return v; }
        if (t == "uchar" || t == "uint8") { uint8_t v; std::memcpy(&v, p, 1); return v; }
        if (t == "char" || t == "int8") { int8_t v; std::memcpy(&v, p, 1); return v; }
        if (t == "ushort" || t == "uint16") { uint16_t v; std::memcpy(&v, p, 2); return v; }
        if (t == "short" || t == "int16") { int16_t v; std::memcpy(&v, p, 2); return v; }
        if (t == "uint" || t == "uint32") { uint32_t v; std::memcpy(&v, p, 4); return v; }
        int32_t v; std::memcpy(&v, p, 4); return v;
    }
    static double decode_pcd(const char* p, char type, size_t size) {  // TYPE F / I / U with SIZE 1, 2, 4, 8
        if (type == 'F') return decode(p, size == 8 ? "double" : "float");
        if (type == 'U') return decode(p, size == 1 ? "uchar" : size == 2 ? "ushort" : "uint");
        if (size == 8) { int64_t v; std::memcpy(&v, p, 8); return (double)v; }
        return decode(p, size == 1 ? "char" : size == 2 ? "short" : "int");
    }
    static void store(PointCloudCPU& c, size_t v, const std::vector<double>& vals, int ix, int iy, int iz, int ii,
                      const int (&rgb3)[3]) {
        (*c.points)[v] = PointType((float)vals[ix], (float)vals[iy], (float)vals[iz], 1.0f);
        if (ii >= 0) (*c.intensities)[v] = (float)vals[ii];
        if (rgb3[0] >= 0)
            (*c.rgb)[v] = RGBType((float)vals[rgb3[0]] / 255.f, (float)vals[rgb3[1]] / 255.f, (float)vals[rgb3[2]] / 255.f, 1.0f);
    }
};

/// io/point_cloud_writer.hpp:14-400. Works on PointCloudCPU and PointCloudShared alike (both expose points / rgb /
/// intensities as host-indexable containers; a PointCloudShared is synchronised to the host once by its accessor).
class PointCloudWriter {
public:
    template <typename PointCloud>
    static void writeFile(const std::string& filename, const PointCloud& cloud, bool binary = false) {
        if (cloud.size() == 0) throw std::runtime_error("[PointCloudWriter::writeFile] Cannot write empty point cloud");
        std::ofstream file(filename, binary ? (std::ios::out | std::ios::binary) : std::ios::out);
        if (!file.is_open())
            throw std::runtime_error("[PointCloudWriter::writeFile] Failed to open file for writing: " + filename);
        const auto dot = filename.find_last_of('.');
        std::string ext = dot == std::string::npos ? "" : filename.substr(dot + 1);
        for (auto& c : ext) c = (char)std::tolower((unsigned char)c);
        if (ext == "ply") write_ply(file, cloud, binary);
        else if (ext == "pcd") write_pcd(file, cloud, binary);
        else throw std::runtime_error("[PointCloudWriter::writeFile] Unsupported file format: " + ext);
        file.close();
        if (file.fail()) throw std::runtime_error("[PointCloudWriter::writeFile] Failed to close file: " + filename);
    }
    template <typename PointCloud>
    static void writePLY(const std::string& filename, const PointCloud& cloud, bool binary = false) {
        writeFile(filename + (filename.find(".ply") == std::string::npos ? ".ply" : ""), cloud, binary);
    }
    template <typename PointCloud>
    static void writePCD(const std::string& filename, const PointCloud& cloud, bool binary = false) {
        writeFile(filename + (filename.find(".pcd") == std::string::npos ? ".pcd" : ""), cloud, binary);
    }

private:
    static bool valid(const PointType& p) { return std::isfinite(p.x()) && std::isfinite(p.y()) && std::isfinite(p.z()); }
    template <typename PointCloud>
    static size_t count_valid(const PointCloud& cloud) {
        size_t n = 0;
        for (size_t i = 0; i < cloud.size(); ++i) n += valid((*cloud.points)[i]) ? 1 : 0;
        return n;
    }
    static float clamp01(float v) { return std::clamp(v, 0.f, 1.f); }

    template <typename PointCloud>
    static void write_ply(std::ofstream& file, const PointCloud& cloud, bool binary) {  // point_cloud_writer.hpp:58-166
        const size_t N = cloud.size();
        const bool has_rgb = cloud.has_rgb(), has_intensity = cloud.has_intensity();
        const size_t n_valid = count_valid(cloud);
        if (n_valid == 0) throw std::runtime_error("[PointCloudWriter::writePLY] No valid points to write");
        file << "ply\n" << (binary ? "format binary_little_endian 1.0\n" : "format ascii 1.0\n");
        file << "element vertex " << n_valid << "\n";
        file << "property float x\nproperty float y\nproperty float z\n";
        if (has_rgb) file << "property uchar red\nproperty uchar green\nproperty uchar blue\n";
        if (has_intensity) file << "property float intensity\n";
        file << "end_header\n";
        if (file.fail()) throw std::runtime_error("[PointCloudWriter::writePLY] Failed to write PLY header");
        if (!binary) file << std::fixed << std::setprecision(6);
        for (size_t i = 0; i < N; ++i) {
            const PointType p = (*cloud.points)[i];
            if (!valid(p)) continue;
            if (binary) {
                const float xyz[3] = {p.x(), p.y(), p.z()};
                file.write(reinterpret_cast<const char*>(xyz), sizeof(xyz));
                if (has_rgb) {
                    const RGBType c = (*cloud.rgb)[i];
                    const uint8_t rgb[3] = {(uint8_t)(clamp01(c.x()) * 255.f), (uint8_t)(clamp01(c.y()) * 255.f),
                                            (uint8_t)(clamp01(c.z()) * 255.f)};
                    file.write(reinterpret_cast<const char*>(rgb), sizeof(rgb));
                }
                if (has_intensity) {
                    const float v = (*cloud.intensities)[i];
                    file.write(reinterpret_cast<const char*>(&v), sizeof(v));
                }
            } else {
                file << p.x() << " " << p.y() << " " << p.z();
                if (has_rgb) {
                    const RGBType c = (*cloud.rgb)[i];
                    file << " " << (int)(clamp01(c.x()) * 255.f) << " " << (int)(clamp01(c.y()) * 255.f) << " "
                         << (int)(clamp01(c.z()) * 255.f);
                }
                if (has_intensity) file << " " << (*cloud.intensities)[i];
                file << "\n";
            }
        }
        if (file.fail()) throw std::runtime_error("[PointCloudWriter::writePLY] Failed to write PLY data");
    }

    template <typename PointCloud>
    static void write_pcd(std::ofstream& file, const PointCloud& cloud, bool binary) {  // point_cloud_writer.hpp:168-290
        const size_t N = cloud.size();
        const bool has_rgb = cloud.has_rgb();
        const size_t n_valid = count_valid(cloud);
        if (n_valid == 0) throw std::runtime_error("[PointCloudWriter::writePCD] No valid points to write");
        file << "# .PCD v.7 - Point Cloud Data file format\nVERSION .7\n";
        file << "FIELDS x y z" << (has_rgb ? " rgb" : "") << "\n";
        file << "SIZE 4 4 4" << (has_rgb ? " 4" : "") << "\n";
        file << "TYPE F F F" << (has_rgb ? " U" : "") << "\n";
        file << "COUNT 1 1 1" << (has_rgb ? " 1" : "") << "\n";
        file << "WIDTH " << n_valid << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << n_valid << "\n";
        file << (binary ? "DATA binary\n" : "DATA ascii\n");
        if (file.fail()) throw std::runtime_error("[PointCloudWriter::writePCD] Failed to write PCD header");
        if (!binary) file << std::fixed << std::setprecision(6);
        for (size_t i = 0; i < N; ++i) {
            const PointType p = (*cloud.points)[i];
            if (!valid(p)) continue;
            int32_t rgb = 0;
            if (has_rgb) {  // one packed word 0x00RRGGBB, no clamping (point_cloud_writer.hpp:244-248)
                const RGBType c = (*cloud.rgb)[i];
                rgb = (int32_t)(((uint32_t)(c.x() * 255.0f) << 16) | ((uint32_t)(c.y() * 255.0f) << 8) | (uint32_t)(c.z() * 255.0f));
            }
            if (binary) {
                const float xyz[3] = {p.x(), p.y(), p.z()};
                file.write(reinterpret_cast<const char*>(xyz), sizeof(xyz));
                if (has_rgb) file.write(reinterpret_cast<const char*>(&rgb), sizeof(rgb));
            } else {
                file << p.x() << " " << p.y() << " " << p.z();
                if (has_rgb) file << " " << rgb;
                file << "\n";
            }
        }
        if (file.fail()) throw std::runtime_error("[PointCloudWriter::writePCD] Failed to write PCD data");
    }
};

}  // namespace sycl_points
