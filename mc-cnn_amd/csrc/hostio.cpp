// Host side of libadcensus that `main.lua` reaches through the same table: adcensus.readPNG16 / writePNG16 / writePFM
// (adcensus.cu:1670-1721; callers main.lua:1212,1218 and the KITTI / Middlebury submission paths) and adcensus.grey2jet
// (adcensus.cu:2000-2053; the debug images of main.lua:503,1242,1260).  Plain host code on HOST
// pointers (the reference takes torch.FloatTensor here, not CudaTensor); no device, no stream.
//
// The reference goes through png++ / libpng; neither has headers in this image, zlib has: the 16-bit greyscale PNG codec
// below is written against the PNG specification (signature, IHDR / IDAT / IEND chunks with CRC-32, zlib stream, the five
// scanline filters) and handles exactly what these functions exchange -- 16-bit greyscale, non-interlaced (KITTI ground truth
// and submission files); 8-bit greyscale files are accepted on read and scaled as libpng's 8 -> 16 bit expansion does (v * 257).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <exception>
#include <new>
#include <vector>
#include <zlib.h>

#include "../../include/mc_adcensus.h"

namespace mc {
void set_error(const char *fmt, ...);
}
using mc::set_error;

namespace {

struct File {
	FILE *f;
	explicit File(const char *name, const char *mode) : f(fopen(name, mode)) {}
	~File() { if (f) fclose(f); }
	bool close() { FILE *g = f; f = nullptr; return g && fclose(g) == 0; }   // (the write paths: a failed flush -- a full disk -- is a failed write)
	long remaining()   // bytes from the current position to the end (-1: not seekable)
	{
		const long at = ftell(f);
		if (at < 0 || fseek(f, 0, SEEK_END)) return -1;
		const long end = ftell(f);
		return (end < 0 || fseek(f, at, SEEK_SET)) ? -1 : end - at;
	}
};

uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
void put_be32(unsigned char *p, uint32_t v) { p[0] = (unsigned char)(v >> 24); p[1] = (unsigned char)(v >> 16); p[2] = (unsigned char)(v >> 8); p[3] = (unsigned char)v; }

const unsigned char PNG_SIG[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};

int paeth(int a, int b, int c)
{
	const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
	return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

bool write_chunk(FILE *f, const char type[4], const unsigned char *data, size_t n)
{
	unsigned char hd[8];
	put_be32(hd, (uint32_t)n);
	memcpy(hd + 4, type, 4);
	uLong crc = crc32(0L, (const Bytef *)type, 4);
	if (n) crc = crc32(crc, data, (uInt)n);
	unsigned char tail[4];
	put_be32(tail, (uint32_t)crc);
	return fwrite(hd, 1, 8, f) == 8 && (n == 0 || fwrite(data, 1, n, f) == n) && fwrite(tail, 1, 4, f) == 4;
}

}  // namespace

namespace {
enum JetForm { JET_ZERO, JET_ONE, JET_UP, JET_DOWN };
struct JetChannel { JetForm form; double a; };
struct JetPiece { double lo, hi; bool hi_closed; JetChannel ch[3]; };   // lo <= val < hi (the last piece: <= hi)
const JetPiece JET[5] = {
	{-0.1, 0.5, false, {{JET_ZERO, 0}, {JET_ZERO, 0}, {JET_UP, -0.5}}},     // adcensus.cu:2022-2025
	{0.5, 1.5, false, {{JET_ZERO, 0}, {JET_UP, 0.5}, {JET_ONE, 0}}},        // :2026-2029
	{1.5, 2.5, false, {{JET_UP, 1.5}, {JET_ONE, 0}, {JET_DOWN, 1.5}}},      // :2030-2033
	{2.5, 3.5, false, {{JET_ONE, 0}, {JET_DOWN, 2.5}, {JET_ZERO, 0}}},      // :2034-2037
	{3.5, 4.1, true, {{JET_DOWN, 3.5}, {JET_ZERO, 0}, {JET_ZERO, 0}}},      // :2038-2041
};
double jet_value(const JetChannel &c, double val)
{
	switch (c.form) {
	case JET_ONE: return 1;
	case JET_UP: return val - c.a;
	case JET_DOWN: return 1 - (val - c.a);
	default: return 0;
	}
}
}  // namespace

extern "C" {

// adcensus.readPNG16(img, fname), adcensus.cu:1670-1686: img[i * width + j] = val == 0 ? 0.0 : val / 256.0 (float).
static int read_png16_impl(const char *fname, float *img, int64_t capacity, int *height, int *width)
{
	if (!fname || !height || !width) { set_error("mc_read_png16: null argument"); return MC_EINVAL; }
	File fp(fname, "rb");
	if (!fp.f) { set_error("mc_read_png16: cannot open %s", fname); return MC_EINVAL; }
	unsigned char sig[8];
	if (fread(sig, 1, 8, fp.f) != 8 || memcmp(sig, PNG_SIG, 8)) { set_error("mc_read_png16: %s is not a PNG file", fname); return MC_EINVAL; }
	uint32_t W = 0, H = 0;
	int depth = 0;
	bool have_hdr = false, done = false;
	std::vector<unsigned char> z;
	while (!done) {
		unsigned char hd[8];
		if (fread(hd, 1, 8, fp.f) != 8) { set_error("mc_read_png16: %s: truncated", fname); return MC_EINVAL; }
		const uint32_t n = be32(hd);
		// (a chunk cannot be longer than what is left of the file: nothing is allocated for a length a damaged file merely claims)
		const long left = fp.remaining();
		// (not seekable -- a pipe: the length cannot be checked against the file, so it is bounded by what a chunk of a 16-bit grey image plausibly
		// holds: encoders write IDAT pieces of 8 KB .. 2 MB; 64 MB is beyond any of them)
		const uint32_t cap = left >= 0 ? (1u << 30) : (64u << 20);
		if (n > cap || (left >= 0 && (unsigned long)left < (unsigned long)n + 4)) {
			set_error("mc_read_png16: %s: %s", fname, n > cap ? "bad chunk length" : "truncated chunk");
			return MC_EINVAL;
		}
		std::vector<unsigned char> d(n + 4);
		if (fread(d.data(), 1, n + 4, fp.f) != n + 4) { set_error("mc_read_png16: %s: truncated chunk", fname); return MC_EINVAL; }
		uLong crc = crc32(0L, hd + 4, 4);
		if (n) crc = crc32(crc, d.data(), n);
		char type[5] = {0, 0, 0, 0, 0};   // (printable form of the chunk type for messages: a damaged file may hold anything there)
		for (int k = 0; k < 4; ++k) type[k] = ((hd[4 + k] | 0x20) >= 'a' && (hd[4 + k] | 0x20) <= 'z') ? (char)hd[4 + k] : '?';
		if ((uint32_t)crc != be32(d.data() + n)) { set_error("mc_read_png16: %s: CRC mismatch in chunk %s", fname, type); return MC_EINVAL; }
		if (!memcmp(hd + 4, "IHDR", 4)) {
			if (n != 13) { set_error("mc_read_png16: %s: bad IHDR", fname); return MC_EINVAL; }
			W = be32(d.data()); H = be32(d.data() + 4);
			depth = d[8];
			const int colour = d[9], interlace = d[12];
			if (colour != 0 || (depth != 16 && depth != 8) || interlace != 0 || d[10] != 0 || d[11] != 0) {
				set_error("mc_read_png16: %s: only non-interlaced 8 / 16-bit greyscale PNGs are supported (colour type %d, depth %d, interlace %d)",
				          fname, colour, depth, interlace);
				return MC_EINVAL;
			}
			if (W == 0 || H == 0 || W > (1u << 20) || H > (1u << 20)) { set_error("mc_read_png16: %s: bad size", fname); return MC_EINVAL; }
			have_hdr = true;
		} else if (!memcmp(hd + 4, "IDAT", 4)) {
			z.insert(z.end(), d.begin(), d.begin() + n);
		} else if (!memcmp(hd + 4, "IEND", 4)) {
			done = true;
		} else if (!(hd[4] & 0x20)) {   // an unknown CRITICAL chunk (upper-case first letter)
			set_error("mc_read_png16: %s: unsupported critical chunk %s", fname, type);
			return MC_EINVAL;
		}
	}
	if (!have_hdr || z.empty()) { set_error("mc_read_png16: %s: no image data", fname); return MC_EINVAL; }
	*height = (int)H; *width = (int)W;
	if (!img) return 0;   // (size query)
	if (capacity < (int64_t)H * W) { set_error("mc_read_png16: %s is %u x %u, the buffer holds %lld pixels", fname, H, W, (long long)capacity); return MC_EINVAL; }
	const size_t bpp = depth / 8, stride = (size_t)W * bpp;
	std::vector<unsigned char> raw((stride + 1) * H);
	uLongf rawlen = (uLongf)raw.size();
	if (uncompress(raw.data(), &rawlen, z.data(), (uLong)z.size()) != Z_OK || rawlen != raw.size()) {
		set_error("mc_read_png16: %s: bad zlib stream", fname);
		return MC_EINVAL;
	}
	std::vector<unsigned char> prev(stride, 0);
	for (uint32_t y = 0; y < H; ++y) {
		unsigned char *row = raw.data() + (stride + 1) * y;
		const int ft = row[0];
		unsigned char *cur = row + 1;
		if (ft > 4) { set_error("mc_read_png16: %s: bad filter type %d", fname, ft); return MC_EINVAL; }
		for (size_t i = 0; i < stride; ++i) {
			const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
			const int pred = ft == 0 ? 0 : ft == 1 ? a : ft == 2 ? b : ft == 3 ? (a + b) / 2 : paeth(a, b, c);
			cur[i] = (unsigned char)(cur[i] + pred);
		}
		for (uint32_t x = 0; x < W; ++x) {
			const uint16_t val = depth == 16 ? (uint16_t)((cur[2 * x] << 8) | cur[2 * x + 1]) : (uint16_t)(cur[x] * 257);
			img[(size_t)y * W + x] = val == 0 ? 0.0f : ((float)val) / 256.0f;   // adcensus.cu:1682
		}
		memcpy(prev.data(), cur, stride);
	}
	return 0;
}

// adcensus.writePNG16(img, height, width, fname), adcensus.cu:1688-1704: (uint16_t)(val < 1e-5 ? 0 : val * 256) per pixel (the
// comparison in double, the product in float, truncation), 16-bit greyscale.
static int write_png16_impl(const float *img, int height, int width, const char *fname)
{
	if (!img || !fname || height < 1 || width < 1) { set_error("mc_write_png16: bad argument"); return MC_EINVAL; }
	const size_t stride = (size_t)width * 2;
	std::vector<unsigned char> raw((stride + 1) * height);
	for (int y = 0; y < height; ++y) {
		unsigned char *row = raw.data() + (stride + 1) * y;
		row[0] = 0;   // filter type None
		for (int x = 0; x < width; ++x) {
			const float val = img[(size_t)y * width + x];
			const float q = (double)val < 1e-5 ? 0.0f : val * 256;
			const uint16_t v = q == q ? (uint16_t)(long long)q : 0;   // (the reference's float -> uint16_t conversion; NaN, which it leaves undefined, becomes 0)
			row[1 + 2 * x] = (unsigned char)(v >> 8);
			row[2 + 2 * x] = (unsigned char)v;
		}
	}
	uLongf zlen = compressBound((uLong)raw.size());
	std::vector<unsigned char> z(zlen);
	if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) { set_error("mc_write_png16: deflate failed"); return MC_EINVAL; }
	File fp(fname, "wb");
	if (!fp.f) { set_error("mc_write_png16: cannot open %s", fname); return MC_EINVAL; }
	unsigned char ihdr[13];
	put_be32(ihdr, (uint32_t)width); put_be32(ihdr + 4, (uint32_t)height);
	ihdr[8] = 16; ihdr[9] = 0; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
	const bool ok = fwrite(PNG_SIG, 1, 8, fp.f) == 8 && write_chunk(fp.f, "IHDR", ihdr, 13) && write_chunk(fp.f, "IDAT", z.data(), zlen) &&
	                write_chunk(fp.f, "IEND", nullptr, 0);
	if (!fp.close() || !ok) { set_error("mc_write_png16: write to %s failed", fname); return MC_EINVAL; }
	return 0;
}

// adcensus.writePFM(img, fname), adcensus.cu:1706-1721: "Pf", "width height", scale -0.003922 (negative = little-endian), the
// rows as stored (no flip), raw floats.
int mc_write_pfm(const float *img, int height, int width, const char *fname)
{
	if (!img || !fname || height < 1 || width < 1) { set_error("mc_write_pfm: bad argument"); return MC_EINVAL; }
	File fp(fname, "wb");
	if (!fp.f) { set_error("mc_write_pfm: cannot open %s", fname); return MC_EINVAL; }
	const bool ok = fprintf(fp.f, "Pf\n%d %d\n-0.003922\n", width, height) >= 0 && fwrite(img, 4, (size_t)height * width, fp.f) == (size_t)height * width;
	if (!fp.close() || !ok) {
		set_error("mc_write_pfm: write to %s failed", fname);
		return MC_EINVAL;
	}
	return 0;
}

// (nothing C++ may leave through the C boundary: an allocation that fails -- the sizes come from a file -- is an error code, not an abort of the
// LuaJIT / ctypes host)
int mc_read_png16(const char *fname, float *img, int64_t capacity, int *height, int *width)
{
	try { return read_png16_impl(fname, img, capacity, height, width); }
	catch (const std::bad_alloc &) { set_error("mc_read_png16: out of memory reading %s", fname ? fname : "(null)"); return MC_EINVAL; }
	catch (const std::exception &e) { set_error("mc_read_png16: %s: %s", fname ? fname : "(null)", e.what()); return MC_EINVAL; }
	catch (...) { set_error("mc_read_png16: %s: unknown exception", fname ? fname : "(null)"); return MC_EINVAL; }
}
int mc_write_png16(const float *img, int height, int width, const char *fname)
{
	try { return write_png16_impl(img, height, width, fname); }
	catch (const std::bad_alloc &) { set_error("mc_write_png16: out of memory writing %s", fname ? fname : "(null)"); return MC_EINVAL; }
	catch (const std::exception &e) { set_error("mc_write_png16: %s: %s", fname ? fname : "(null)", e.what()); return MC_EINVAL; }
	catch (...) { set_error("mc_write_png16: %s: unknown exception", fname ? fname : "(null)"); return MC_EINVAL; }
}

// adcensus.grey2jet(grey_img, col_img), adcensus.cu:2000-2053: the jet colour map over val = 4 * grey, doubles, planes red / green / blue.
// The map is five pieces of val, on each of which a channel is 0, 1, a ramp up `val - a` or a ramp down `1 - (val - a)` -- held here as a
// table and evaluated in the reference's expressions (its `0.5 + val` of the first piece is `val - (-0.5)`: the same double).  The
// reference asserts on a value outside [-0.1, 4.1]; here: an error with the pixel named, the pixels before it written as there.

int mc_grey2jet(const double *grey, double *col, int height, int width)
{
	if (!grey || !col || height < 1 || width < 1) { set_error("mc_grey2jet: bad argument"); return MC_EINVAL; }
	const size_t hw = (size_t)height * width;
	for (size_t px = 0; px < hw; ++px) {
		const double val = grey[px] * 4;
		const JetPiece *piece = nullptr;
		for (const JetPiece &q : JET)
			if (q.lo <= val && (q.hi_closed ? val <= q.hi : val < q.hi)) { piece = &q; break; }
		if (!piece) {   // (NaN fails every comparison)
			set_error("mc_grey2jet: val = %f at (%d, %d) is outside the colour map's range [-0.1, 4.1]", val, (int)(px / width), (int)(px % width));
			return MC_EINVAL;
		}
		for (int c = 0; c < 3; ++c) col[c * hw + px] = jet_value(piece->ch[c], val);
	}
	return 0;
}

}  // extern "C"
