// sycl_points facade for MI355X — minimal PLY reader (io/point_cloud_reader.hpp:20-276, 494-546).
// Reads `element vertex` with scalar properties from ascii or binary_little_endian files: x, y, z (any scalar type) into
// PointCloudCPU::points (w = 1); `intensity` / `scalar_intensity` into intensities when asked. Everything else of the
// reference's I/O layer (PCD, writers, rgb packing) is outside the hot path's scope.
#pragma once
#include <cstdio>
#include <fstream>
#include <sstream>

#include "core.hpp"

namespace sycl_points {

class PointCloudReader {
public:
    static PointCloudCPU readFile(const std::string& filename, bool read_rgb = false, bool read_intensity = false) {
        const auto dot = filename.find_last_of('.');
        std::string ext = dot == std::string::npos ? "" : filename.substr(dot + 1);
        for (auto& c : ext) c = (char)std::tolower((unsigned char)c);
        (void)read_rgb;
        if (ext == "ply") return readPLY(filename, read_intensity);
        if (ext == "pcd") return readPCD(filename, read_intensity);
        throw std::runtime_error("[PointCloudReader::readFile] Unsupported file format: " + filename);
    }

    /// PCD (io/point_cloud_reader.hpp:278-492): header FIELDS / SIZE / TYPE / COUNT / WIDTH / HEIGHT / POINTS / DATA,
    /// `DATA ascii` or `DATA binary` (little endian rows of the declared fields). x, y, z are required; `intensity`
    /// is read when asked for. `binary_compressed` is rejected (as in the reference, which only knows ascii and binary).
    static PointCloudCPU readPCD(const std::string& filename, bool read_intensity = false) {
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
        int ix = -1, iy = -1, iz = -1, ii = -1;
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
        }
        if (ix < 0 || iy < 0 || iz < 0) throw std::runtime_error("[PointCloudReader::readPCD] x/y/z fields missing");
        PointCloudCPU cloud;
        cloud.points->resize(n_points);
        const bool want_i = read_intensity && ii >= 0;
        if (want_i) cloud.intensities->resize(n_points);
        std::vector<double> vals(n_vals);
        if (data == "ascii") {
            for (size_t v = 0; v < n_points; ++v) {
                for (size_t k = 0; k < n_vals; ++k)
                    if (!(f >> vals[k])) throw std::runtime_error("[PointCloudReader::readPCD] Error reading ascii PCD data");
                store(cloud, v, vals, ix, iy, iz, want_i ? ii : -1);
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
                store(cloud, v, vals, ix, iy, iz, want_i ? ii : -1);
            }
        }
        return cloud;
    }

    static PointCloudCPU readPLY(const std::string& filename, bool read_intensity = false) {
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
        int ix = -1, iy = -1, iz = -1, ii = -1;
        for (size_t i = 0; i < props.size(); ++i) {
            if (props[i].name == "x") ix = (int)i;
            else if (props[i].name == "y") iy = (int)i;
            else if (props[i].name == "z") iz = (int)i;
            else if (props[i].name == "intensity" || props[i].name == "scalar_intensity") ii = (int)i;
        }
        if (ix < 0 || iy < 0 || iz < 0) throw std::runtime_error("[readPLY] x/y/z properties missing");
        PointCloudCPU cloud;
        cloud.points->resize(n_vertex);
        const bool want_i = read_intensity && ii >= 0;
        if (want_i) cloud.intensities->resize(n_vertex);
        std::vector<double> vals(props.size());
        if (format == "ascii") {
            for (size_t v = 0; v < n_vertex; ++v) {
                for (size_t p = 0; p < props.size(); ++p) f >> vals[p];
                store(cloud, v, vals, ix, iy, iz, want_i ? ii : -1);
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
                store(cloud, v, vals, ix, iy, iz, want_i ? ii : -1);
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
    static void store(PointCloudCPU& c, size_t v, const std::vector<double>& vals, int ix, int iy, int iz, int ii) {
        (*c.points)[v] = PointType((float)vals[ix], (float)vals[iy], (float)vals[iz], 1.0f);
        if (ii >= 0) (*c.intensities)[v] = (float)vals[ii];
    }
};

}  // namespace sycl_points
