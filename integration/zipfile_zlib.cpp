// integration/zipfile_zlib.cpp — `ZipFile` of KataGo's cpp/dataio/numpywrite.h:45-61 without libzip.
//
// The reference writes its self-play training data as .npz files: a zip archive whose members are the .npy images that
// `NumpyBuffer<T>` prepares (`TrainingWriteBuffers::writeToZipFile`, cpp/dataio/trainingwrite.cpp:854-886). Its only zip
// implementation is a thin wrapper over libzip (cpp/dataio/numpywrite.cpp:266-322); built with NO_LIBZIP every member
// function throws (:240-264), so `selfplay` — the command BASELINE's games/hour metric is defined on — cannot write data
// on a machine without libzip. This file defines the same four member functions over zlib, which KataGo links anyway
// (model files are gzip): one local header + raw-deflate stream per member, central directory and end record at close().
//
// Behaviour kept from the libzip wrapper:
//   * constructor creates/truncates the file and throws StringError when it cannot;
//   * writeBuffer(name, data, n) adds one member called exactly `name` (no ".npy" suffix is added — numpy's NpzFile
//     accepts either), deflate-compressed; a second write under the same name replaces the first (ZIP_FL_OVERWRITE);
//   * nothing is a valid archive until close(); the destructor of an unclosed ZipFile discards (zip_discard) — here the
//     partial file is removed;
//   * errors are StringError with the file name in the message.
// Difference: libzip reads `data` lazily at close(); this implementation compresses during writeBuffer, so the caller's
// buffer may be reused immediately (a weaker requirement, compatible with every caller).
// Limits: members and archives up to 4 GiB − 1 (no zip64): the reference's largest file is maxRowsPerTrainFile rows of
// ≈ 3.3 KB compressed, far below; exceeding it throws instead of writing a corrupt archive.
//
// Build: compile the reference's dataio/numpywrite.cpp with -DNO_LIBZIP -DZipFile=ZipFileWithoutLibzip (its throwing stub
// then gets another name) and link this file; or, in the KataGo tree, replace the NO_LIBZIP branch with it.

#include "dataio/numpywrite.h"

#include <zlib.h>

#include <cerrno>
#include <cstdio>
#include <cstring>

using namespace std;

namespace {

struct Member {
  string name;
  uint32_t crc;
  uint32_t compressedSize;
  uint32_t uncompressedSize;
  uint32_t localHeaderOffset;
  bool live;  // false once a later member of the same name replaced it
};

struct Archive {
  FILE* fp;
  vector<Member> members;
  uint64_t offset;
};

const uint16_t VERSION_NEEDED = 20;   // 2.0: deflate
const uint16_t METHOD_DEFLATE = 8;
const uint16_t DOS_TIME = 0;          // 00:00:00
const uint16_t DOS_DATE = (1 << 5) | 1;  // 1980-01-01: archives are reproducible byte for byte

void put16(vector<unsigned char>& v, uint16_t x) {
  v.push_back((unsigned char)(x & 0xff));
  v.push_back((unsigned char)(x >> 8));
}
void put32(vector<unsigned char>& v, uint32_t x) {
  put16(v, (uint16_t)(x & 0xffff));
  put16(v, (uint16_t)(x >> 16));
}

void writeAll(Archive* ar, const string& fileName, const void* p, size_t n) {
  if(n > 0 && fwrite(p, 1, n, ar->fp) != n)
    throw StringError("Could not write to zip file " + fileName + ": " + strerror(errno));
  ar->offset += n;
  if(ar->offset > 0xfffffffeULL)
    throw StringError("Zip file " + fileName + " would exceed 4 GiB, which this writer does not support");
}

}  // namespace

ZipFile::ZipFile(const string& fName)
  :fileName(fName),file(NULL)
{
  FILE* fp = fopen(fileName.c_str(), "wb");
  if(fp == NULL)
    throw StringError("Could not open zip file " + fileName + " due to error " + strerror(errno));
  Archive* ar = new Archive();
  ar->fp = fp;
  ar->offset = 0;
  file = ar;
}

ZipFile::~ZipFile() {
  if(file != NULL) {
    Archive* ar = (Archive*)file;
    fclose(ar->fp);
    remove(fileName.c_str());
    delete ar;
  }
}

void ZipFile::writeBuffer(const char* nameWithinZip, void* data, uint64_t numBytes) {
  Archive* ar = (Archive*)file;
  if(ar == NULL)
    throw StringError("Could not write to " + string(nameWithinZip) + " within zip file " + fileName + ": file is closed");
  if(numBytes > 0xfffffffeULL)
    throw StringError("Could not write to " + string(nameWithinZip) + " within zip file " + fileName + ": member of 4 GiB or more");
  const string name(nameWithinZip);
  if(name.size() == 0 || name.size() > 0xffff)
    throw StringError("Invalid member name within zip file " + fileName);

  Member m;
  m.name = name;
  m.crc = 0;
  m.uncompressedSize = (uint32_t)numBytes;
  m.localHeaderOffset = (uint32_t)ar->offset;
  m.live = true;
  {
    // crc32 takes 32-bit lengths
    uLong crc = crc32(0L, Z_NULL, 0);
    const unsigned char* p = (const unsigned char*)data;
    uint64_t left = numBytes;
    while(left > 0) {
      uInt chunk = (uInt)(left > (1u << 30) ? (1u << 30) : left);
      crc = crc32(crc, p, chunk);
      p += chunk;
      left -= chunk;
    }
    m.crc = (uint32_t)crc;
  }

  // Raw deflate into memory first: the local header carries the sizes, so no data descriptor is needed and strict
  // readers (and Python's zipfile) see ordinary members.
  vector<unsigned char> comp;
  {
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if(deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK)
      throw StringError("Could not initialize zip write data buffer for " + name + " within " + fileName);
    comp.resize(deflateBound(&zs, (uLong)numBytes) + 64);
    const unsigned char* p = (const unsigned char*)data;
    uint64_t left = numBytes;
    size_t produced = 0;
    int ret = Z_OK;
    do {
      uInt chunk = (uInt)(left > (1u << 30) ? (1u << 30) : left);
      zs.next_in = const_cast<Bytef*>(p);
      zs.avail_in = chunk;
      p += chunk;
      left -= chunk;
      const int flush = left == 0 ? Z_FINISH : Z_NO_FLUSH;
      do {
        if(produced == comp.size())
          comp.resize(comp.size() * 2);
        size_t room = comp.size() - produced;
        zs.next_out = comp.data() + produced;
        zs.avail_out = (uInt)(room > (1u << 30) ? (1u << 30) : room);
        const uInt before = zs.avail_out;
        ret = deflate(&zs, flush);
        if(ret == Z_STREAM_ERROR) {
          deflateEnd(&zs);
          throw StringError("Could not write to " + name + " within zip file " + fileName + " due to a deflate error");
        }
        produced += before - zs.avail_out;
      } while(zs.avail_in > 0 || (flush == Z_FINISH && ret != Z_STREAM_END));
    } while(left > 0);
    deflateEnd(&zs);
    if(ret != Z_STREAM_END)
      throw StringError("Could not write to " + name + " within zip file " + fileName + " due to a deflate error");
    comp.resize(produced);
  }
  if(comp.size() > 0xfffffffeULL)
    throw StringError("Could not write to " + name + " within zip file " + fileName + ": member of 4 GiB or more");
  m.compressedSize = (uint32_t)comp.size();

  vector<unsigned char> h;
  put32(h, 0x04034b50u);
  put16(h, VERSION_NEEDED);
  put16(h, 0);  // flags
  put16(h, METHOD_DEFLATE);
  put16(h, DOS_TIME);
  put16(h, DOS_DATE);
  put32(h, m.crc);
  put32(h, m.compressedSize);
  put32(h, m.uncompressedSize);
  put16(h, (uint16_t)name.size());
  put16(h, 0);  // extra
  h.insert(h.end(), name.begin(), name.end());
  writeAll(ar, fileName, h.data(), h.size());
  writeAll(ar, fileName, comp.data(), comp.size());

  for(Member& old : ar->members)
    if(old.live && old.name == name)
      old.live = false;  // ZIP_FL_OVERWRITE: only the newest is listed in the central directory
  ar->members.push_back(m);
}

void ZipFile::close() {
  Archive* ar = (Archive*)file;
  if(ar == NULL)
    throw StringError("Could not close zip file " + fileName + ": already closed");
  const uint32_t dirOffset = (uint32_t)ar->offset;
  uint32_t count = 0;
  vector<unsigned char> dir;
  for(const Member& m : ar->members) {
    if(!m.live)
      continue;
    put32(dir, 0x02014b50u);
    put16(dir, VERSION_NEEDED);  // made by: 2.0, MS-DOS attribute convention
    put16(dir, VERSION_NEEDED);
    put16(dir, 0);
    put16(dir, METHOD_DEFLATE);
    put16(dir, DOS_TIME);
    put16(dir, DOS_DATE);
    put32(dir, m.crc);
    put32(dir, m.compressedSize);
    put32(dir, m.uncompressedSize);
    put16(dir, (uint16_t)m.name.size());
    put16(dir, 0);  // extra
    put16(dir, 0);  // comment
    put16(dir, 0);  // disk number
    put16(dir, 0);  // internal attributes
    put32(dir, 0);  // external attributes
    put32(dir, m.localHeaderOffset);
    dir.insert(dir.end(), m.name.begin(), m.name.end());
    count++;
  }
  if(count > 0xfffe)
    throw StringError("Could not close zip file " + fileName + ": too many members");
  writeAll(ar, fileName, dir.data(), dir.size());
  vector<unsigned char> end;
  put32(end, 0x06054b50u);
  put16(end, 0);
  put16(end, 0);
  put16(end, (uint16_t)count);
  put16(end, (uint16_t)count);
  put32(end, (uint32_t)dir.size());
  put32(end, dirOffset);
  put16(end, 0);
  writeAll(ar, fileName, end.data(), end.size());
  const bool flushed = fflush(ar->fp) == 0;
  const bool closed = fclose(ar->fp) == 0;
  delete ar;
  file = NULL;
  if(!flushed || !closed)
    throw StringError("Could not close zip file " + fileName + " due to error " + strerror(errno));
}
