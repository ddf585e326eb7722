// x3 -- the reference's command line (src/bin/x3.rs:43-82) on top of libx3hip.so:
//   x3 --input a.wav --output a.x3a      encode (encodefile::wav_to_x3a)
//   x3 -i a.x3a -o a.wav                 decode (decodefile::x3a_to_wav)
// Same two options, the same rule for telling the direction from the file extensions, the same
// statistics block after an encode.  Where the reference panics (bad extension, same type on both
// sides, a failing conversion's .unwrap()) this prints the reason and exits with status 101, the exit
// status of a Rust panic.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../host/x3.hpp"

enum class AudioFile { X3a, Wav, Invalid };

static bool ends_with(const std::string& s, const char* suffix) {
  const size_t n = std::strlen(suffix);
  return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}
static AudioFile filetype(const std::string& name) {  // get_filetype (bin/x3.rs:33-41)
  if (ends_with(name, ".x3a")) return AudioFile::X3a;
  if (ends_with(name, ".wav")) return AudioFile::Wav;
  return AudioFile::Invalid;
}

static int usage(const char* why) {
  std::fprintf(stderr, "error: %s\n\nUSAGE:\n    x3 --input <FILE> --output <FILE>\n\n"
               "    -i, --input <FILE>     The input file, a .wav or .x3a file\n"
               "    -o, --output <FILE>    The output file, a .wav or .x3a file\n", why);
  return 2;  // clap's exit status for a usage error
}

int main(int argc, char** argv) {
  std::string in_file, out_file;
  int device = 0;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto value = [&](std::string* dst) -> bool {
      if (i + 1 >= argc) return false;
      *dst = argv[++i];
      return true;
    };
    if (a == "-i" || a == "--input") {
      if (!value(&in_file)) return usage("--input needs a value");
    } else if (a == "-o" || a == "--output") {
      if (!value(&out_file)) return usage("--output needs a value");
    } else if (a == "--device") {  // not in the reference: which GPU
      std::string d;
      if (!value(&d)) return usage("--device needs a value");
      device = std::atoi(d.c_str());
    } else if (a == "-V" || a == "--version") {
      std::printf("x3 0.3.0 (libx3hip, MI355X)\n");
      return 0;
    } else {
      return usage(("unexpected argument '" + a + "'").c_str());
    }
  }
  if (in_file.empty() || out_file.empty()) return usage("the following required arguments were not provided: --input <FILE> --output <FILE>");
  const AudioFile in_type = filetype(in_file), out_type = filetype(out_file);
  for (const std::string* f : {&in_file, &out_file})
    if (filetype(*f) == AudioFile::Invalid) {
      std::fprintf(stderr, "Invalid audio file, expecting a '.wav' or '.x3a' file: %s\n", f->c_str());
      return 101;
    }
  if (in_type == out_type) {
    std::fprintf(stderr, "Input must be different file type than output.\n");
    return 101;
  }
  x3_ctx* raw = nullptr;
  if (x3_ctx_create(device, &raw) != X3_OK) {
    std::fprintf(stderr, "no usable HIP device %d (there is no CPU fallback)\n", device);
    return 101;
  }
  x3_ctx_destroy(raw);
  x3::Context ctx(device);
  x3::X3Error e;
  if (in_type == AudioFile::Wav) {
    e = x3::encodefile::wav_to_x3a(ctx, in_file.c_str(), out_file.c_str());
  } else {
    uint64_t samples = 0, frame_errors = 0;
    e = x3::decodefile::x3a_to_wav(ctx, in_file.c_str(), out_file.c_str(), &samples, &frame_errors);
    if (frame_errors) std::printf("Frame error: the stream ends at the first frame that does not decode (%llu samples written)\n",
                                  (unsigned long long)samples);
  }
  if (e != x3::X3Error::Ok) {
    std::fprintf(stderr, "called `Result::unwrap()` on an `Err` value: %s (%s)\n", x3::to_string(e), x3_last_error(ctx.raw()));
    return 101;
  }
  return 0;
}
