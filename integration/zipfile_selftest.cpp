// Driver for tests/test_npz_writer.py: exercises integration/zipfile_zlib.cpp on its own (no game code).
//   zipfile_selftest <dir>   writes <dir>/ok.npz-like archive, an abandoned archive and tries an unwritable path;
// prints one line per step; the Python test reads the archive back with zipfile.
#include "dataio/numpywrite.h"

#include <cstdio>
#include <cstdlib>
#include <iostream>

using namespace std;

int main(int argc, char** argv) {
  if(argc != 2) { cerr << "usage: zipfile_selftest DIR" << endl; return 2; }
  const string dir = argv[1];
  {
    ZipFile z(dir + "/ok.zip");
    vector<unsigned char> big(3 * 1000 * 1000 + 17);
    uint32_t s = 12345;
    for(size_t i = 0; i < big.size(); i++) { s = s * 1664525u + 1013904223u; big[i] = (unsigned char)((s >> 24) & ((i / 4096) % 2 ? 0x0f : 0xff)); }
    vector<unsigned char> first(1000, 7), second(2000, 9);
    unsigned char nothing = 0;
    z.writeBuffer("empty", &nothing, 0);
    z.writeBuffer("replaced", first.data(), first.size());
    z.writeBuffer("big", big.data(), big.size());
    for(size_t i = 0; i < first.size(); i++) first[i] = 0;  // the caller's buffer may be reused at once
    z.writeBuffer("replaced", second.data(), second.size());
    NumpyBuffer<float> np({4, 3, 2});
    for(int i = 0; i < 4 * 3 * 2; i++) np.data[i] = 0.5f * i;
    uint64_t nbytes = np.prepareHeaderWithNumRows(3);  // partial batch: 3 of 4 rows
    z.writeBuffer("partial", np.dataIncludingHeader, nbytes);
    z.close();
    cout << "closed ok.zip" << endl;
  }
  {
    ZipFile z(dir + "/abandoned.zip");
    int x = 5;
    z.writeBuffer("x", &x, sizeof(x));
    // no close(): discarded
  }
  cout << "abandoned" << endl;
  try {
    ZipFile z(dir + "/no/such/dir/file.zip");
    cout << "ERROR: opened an unwritable path" << endl;
    return 1;
  }
  catch(const StringError& e) {
    cout << "threw: " << e.what() << endl;
  }
  return 0;
}
