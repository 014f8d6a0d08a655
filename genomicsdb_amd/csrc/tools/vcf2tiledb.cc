// vcf2tiledb - command line of the reference's import tool for what this build implements:
//   vcf2tiledb [-r <rank>] <loader.json>
// (reference: tools/src/vcf2tiledb.cc:54-140; VCF2TileDBLoader::read_all, src/main/cpp/src/loader/tiledb_loader.cc:845-965).
// The (g)VCFs of the callset mapping are converted to begin-cells of column partition `rank` (host, vcf_importer.cc) and
//   "produce_tiledb_array": true  -> written to <workspace>/<array>/cells.bin, the array file gt_mpi_gather and the JNI stream open
//   "produce_combined_vcf": true  -> combined in-line on the GPU (same path as gt_mpi_gather --produce-Broad-GVCF) and written
//                                    to stdout, like the reference loader's in-line BroadCombinedGVCFOperator
// The --split-files modes of the reference tool are not implemented (exit with an error).
#include <getopt.h>
#include <sys/stat.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "../api/genomicsdb_bcf_generator.h"
#include "../common/mini_json.hpp"
#include "../host/vcf_importer.h"

using namespace genomicsdb_amd;

static int launcher_rank() {
  for (const char* name : {"OMPI_COMM_WORLD_RANK", "PMI_RANK", "PMIX_RANK", "SLURM_PROCID", "RANK"})
    if (const char* e = getenv(name)) return atoi(e);
  return 0;
}
static void mkdir_p(const std::string& path) {
  for (size_t i = 1; i <= path.size(); ++i)
    if (i == path.size() || path[i] == '/') mkdir(path.substr(0, i).c_str(), 0755);
}

int main(int argc, char** argv) {
  enum { ARG_VERSION = 1000, ARG_UNSUPPORTED };
  static struct option long_options[] = {{"tmp-directory", 1, 0, 'T'}, {"rank", 1, 0, 'r'}, {"version", 0, 0, ARG_VERSION},
                                         {"split-files", 0, 0, ARG_UNSUPPORTED}, {"split-all-partitions", 0, 0, ARG_UNSUPPORTED},
                                         {"split-files-results-directory", 1, 0, ARG_UNSUPPORTED}, {"split-output-filename", 1, 0, ARG_UNSUPPORTED},
                                         {"split-callset-mapping-file", 0, 0, ARG_UNSUPPORTED}, {0, 0, 0, 0}};
  int rank = launcher_rank();
  int c;
  while ((c = getopt_long(argc, argv, "T:r:", long_options, NULL)) >= 0) {
    switch (c) {
      case 'T': break;   // no temporary files are written
      case 'r': rank = atoi(optarg); break;
      case ARG_VERSION: std::cout << "genomicsdb_amd (MI355X variant-combine path) for GenomicsDB 0.10.2 loader JSON\n"; return 0;
      case ARG_UNSUPPORTED: std::cerr << "vcf2tiledb: the --split-files modes are not implemented by this build\n"; return -1;
      default: std::cerr << "Unknown command line argument\n"; return -1;
    }
  }
  if (optind + 1 > argc) { std::cerr << "Needs 1 argument <loader_json_config_file>\n"; return -1; }
  const std::string loader_json = argv[optind];
  try {
    const auto t0 = std::chrono::steady_clock::now();
    const std::string loader_text = mini_json::read_text_file(loader_json);
    const mini_json::Value doc = mini_json::parse(loader_text);
    GenomicsDBImportConfig loader;
    loader.read_from_json(doc, rank);
    if (loader.m_vid_mapping_file.empty() || loader.m_callset_mapping_file.empty()) throw GenomicsDBConfigException("loader JSON needs vid_mapping_file and callset_mapping_file");
    VidMapper vid;
    vid.parse_vid_json(mini_json::parse_file(loader.m_vid_mapping_file));
    vid.parse_callsets_json(mini_json::parse_file(loader.m_callset_mapping_file));
    const ColumnRange part = loader.get_column_partition(rank);
    ImportOptions opt;
    opt.treat_deletions_as_intervals = loader.m_treat_deletions_as_intervals;
    opt.column_begin = part.first;
    opt.column_end = part.second;
    ImportStats st;
    const std::vector<uint8_t> cells = import_callsets_to_cells(vid, opt, &st);
    const double t_import = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cerr << "GENOMICSDB_TIMER,Rank," << rank << ",vcf2binary,Wall-clock time(s)," << t_import << ",files," << st.num_files << ",records," << st.num_records
              << ",cells," << st.num_cells << ",bytes," << st.num_bytes << "\n";
    const bool produce_array = doc.HasMember("produce_tiledb_array") && doc["produce_tiledb_array"].GetBool();
    if (produce_array) {
      const std::string dir = loader.get_workspace(rank) + "/" + loader.get_array_name(rank);
      mkdir_p(dir);
      FILE* f = fopen((dir + "/cells.bin").c_str(), "wb");
      if (!f) throw GenomicsDBConfigException("cannot write " + dir + "/cells.bin");
      if (!cells.empty() && fwrite(cells.data(), 1, cells.size(), f) != cells.size()) { fclose(f); throw GenomicsDBConfigException("short write to " + dir + "/cells.bin"); }
      fclose(f);
      remove((dir + "/fragment.gdbamd").c_str());   // a columnar copy of older cells must not shadow the new array
      // "compress_tiledb_array" (the reference's loader gzips the attribute tiles of the array it writes): the columnar fragment file with
      // its data sections as DEFLATE tiles goes next to cells.bin; queries read it window by window and inflate the tiles on the device
      if (doc.HasMember("compress_tiledb_array") && doc["compress_tiledb_array"].GetBool() && !cells.empty()) {
        const auto tc = std::chrono::steady_clock::now();
        const size_t close_at = loader_text.rfind('}');
        if (close_at == std::string::npos) throw GenomicsDBConfigException("loader JSON is not an object");
        const std::string all_text = loader_text.substr(0, close_at) + ", \"query_column_ranges\": [[[" + std::to_string(part.first) + ", " + std::to_string(part.second) + "]]]" +
                                     loader_text.substr(close_at);
        try {
          CombineEngine eng(mini_json::parse(all_text), 0, nullptr, rank, "", false);
          eng.stage_cells(cells.data(), cells.size());
          eng.save_fragment(dir + "/fragment.gdbamd", true);
          struct stat fs;
          const long long zbytes = ::stat((dir + "/fragment.gdbamd").c_str(), &fs) == 0 ? (long long)fs.st_size : -1;
          std::cerr << "GENOMICSDB_TIMER,Rank," << rank << ",compress_tiledb_array,Wall-clock time(s)," << std::chrono::duration<double>(std::chrono::steady_clock::now() - tc).count()
                    << ",cells_bytes," << cells.size() << ",fragment_bytes," << zbytes << "\n";
        } catch (const GenomicsDBDeviceException& e) {
          // (the columnar file is built from columns staged in HBM: without a device the array stays as cells.bin, which every reader accepts)
          std::cerr << "vcf2tiledb: compress_tiledb_array skipped: " << e.what() << "\n";
        }
      }
    }
    if (loader.m_produce_combined_vcf) {
      // the loader JSON doubles as the query JSON of the in-line combine: every attribute, the whole column partition
      const size_t close = loader_text.rfind('}');
      if (close == std::string::npos) throw GenomicsDBConfigException("loader JSON is not an object");
      const std::string query_text = loader_text.substr(0, close) + ", \"query_column_ranges\": [[[" + std::to_string(part.first) + ", " + std::to_string(part.second) +
                                     "]]]" + loader_text.substr(close);
      const size_t capacity = (size_t)64u << 20;
      GenomicsDBBCFGenerator gen(query_text, cells.data(), cells.size(), capacity, false);
      std::vector<uint8_t> buf(capacity);
      size_t total = 0;
      while (!gen.end()) {
        const size_t n = gen.read_and_advance(buf.data(), 0, buf.size());
        if (n == 0) break;
        if (fwrite(buf.data(), 1, n, stdout) != n) { std::cerr << "short write\n"; return -1; }
        total += n;
      }
      fflush(stdout);
      const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() - t_import;
      std::cerr << "GENOMICSDB_TIMER,Rank," << rank << ",produce_combined_vcf,Wall-clock time(s)," << secs << ",bytes," << total << "\n";
    }
  } catch (const std::exception& e) {
    std::cerr << "vcf2tiledb: " << e.what() << "\n";
    return -1;
  }
  return 0;
}
