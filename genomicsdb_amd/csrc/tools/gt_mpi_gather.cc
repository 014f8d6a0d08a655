// gt_mpi_gather - command line of the reference's query tool for the one mode this build implements:
//   gt_mpi_gather -j <query.json> [-l <loader.json>] [-r <rank>] [-p <page size>] [-s <segment size>] [-O <fmt>] --produce-Broad-GVCF
// (reference: tools/src/gt_mpi_gather.cc:437-531; scan_and_produce_Broad_GVCF :322-366).  One process per column partition
// like `mpirun -n P gt_mpi_gather`: the rank comes from -r, else from the launcher (OMPI_COMM_WORLD_RANK / PMI_RANK / RANK);
// ranks do not communicate.  The VCF goes to "vcf_output_filename" of the query JSON (per-rank entry if it is a list), else to
// stdout.  The scan + combine runs on the GPU (LOCAL_RANK / GDBAMD_DEVICE selects it); there is no CPU path.
#include <getopt.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "../api/genomicsdb_bcf_generator.h"
#include "../common/mini_json.hpp"
#include "../host/vcf_index.h"

using namespace genomicsdb_amd;

enum { ARGS_IDX_PRODUCE_BROAD_GVCF = 1000, ARGS_IDX_PRODUCE_HISTOGRAM, ARGS_IDX_PRINT_CALLS, ARGS_IDX_PRINT_CSV, ARGS_IDX_PRINT_AC, ARGS_IDX_VERSION, ARGS_IDX_UNSUPPORTED };
extern "C" int64_t gdbamd_equi_partition_text(const uint64_t* counts, uint64_t nbins, uint64_t hist_begin, uint64_t bin_size, uint64_t num_parts, char* dst, uint64_t cap);

static int launcher_rank() {
  for (const char* name : {"OMPI_COMM_WORLD_RANK", "PMI_RANK", "PMIX_RANK", "SLURM_PROCID", "RANK"})
    if (const char* e = getenv(name)) return atoi(e);
  return 0;
}

int main(int argc, char** argv) {
  static struct option long_options[] = {
      {"page-size", 1, 0, 'p'}, {"rank", 1, 0, 'r'}, {"output-format", 1, 0, 'O'}, {"workspace", 1, 0, 'w'}, {"json-config", 1, 0, 'j'},
      {"loader-json-config", 1, 0, 'l'}, {"segment-size", 1, 0, 's'}, {"array", 1, 0, 'A'},
      {"produce-Broad-GVCF", 0, 0, ARGS_IDX_PRODUCE_BROAD_GVCF}, {"version", 0, 0, ARGS_IDX_VERSION},
      {"skip-query-on-root", 0, 0, ARGS_IDX_UNSUPPORTED}, {"produce-interesting-positions", 0, 0, ARGS_IDX_UNSUPPORTED},
      {"produce-histogram", 0, 0, ARGS_IDX_PRODUCE_HISTOGRAM}, {"print-calls", 0, 0, ARGS_IDX_PRINT_CALLS}, {"print-csv", 0, 0, ARGS_IDX_PRINT_CSV},
      {"print-AC", 0, 0, ARGS_IDX_PRINT_AC}, {0, 0, 0, 0}};
  std::string json_config, loader_json, output_format;
  size_t page_size = 0, segment_size = 10u * 1024u * 1024u;
  int rank = launcher_rank();
  bool produce_gvcf = false, produce_histogram = false, print_calls = false;
  int print_mode = 0;
  int c;
  while ((c = getopt_long(argc, argv, "j:l:w:A:p:O:s:r:", long_options, NULL)) >= 0) {
    switch (c) {
      case 'p': page_size = strtoull(optarg, 0, 10); break;
      case 'r': rank = atoi(optarg); break;
      case 'O': output_format = optarg; break;
      case 's': segment_size = strtoull(optarg, 0, 10); break;
      case 'j': json_config = optarg; break;
      case 'l': loader_json = optarg; break;
      case 'w': case 'A': std::cerr << "-w / -A: give workspace and array in the JSON files\n"; return -1;
      case ARGS_IDX_PRODUCE_BROAD_GVCF: produce_gvcf = true; break;
      case ARGS_IDX_PRODUCE_HISTOGRAM: produce_histogram = true; break;
      case ARGS_IDX_PRINT_CALLS: print_calls = true; print_mode = 0; break;
      case ARGS_IDX_PRINT_CSV: print_calls = true; print_mode = 1; break;
      case ARGS_IDX_PRINT_AC: print_calls = true; print_mode = 2; break;
      case ARGS_IDX_VERSION: std::cout << "genomicsdb_amd (MI355X variant-combine path) for GenomicsDB 0.10.2 query JSON\n"; return 0;
      case ARGS_IDX_UNSUPPORTED: std::cerr << "this build implements --produce-Broad-GVCF, --produce-histogram, --print-calls, --print-csv and --print-AC only\n"; return -1;
      default: std::cerr << "Unknown command line argument\n"; return -1;
    }
  }
  if (json_config.empty() || !(produce_gvcf || produce_histogram || print_calls)) {
    std::cerr << "Usage: gt_mpi_gather -j <query.json> [-l <loader.json>] [-r rank] [-p page_size] [-O output_format] --produce-Broad-GVCF | --produce-histogram | --print-calls | --print-csv | --print-AC\n";
    return -1;
  }
  if (print_calls) {
    // print_calls, COMMAND_PRINT_CALLS (tools/src/gt_mpi_gather.cc:369-383): the cells of the query intervals as JSON; selected and formatted on the GPU
    try {
      GenomicsDBBCFGenerator gen(loader_json, json_config, "", 0, 0, rank, (size_t)1u << 20, segment_size, "", true, false, true);
      const std::string doc = print_mode == 0 ? gen.engine().print_calls() : print_mode == 1 ? gen.engine().print_csv() : gen.engine().print_allele_counts();
      fwrite(doc.data(), 1, doc.size(), stdout);
    } catch (const std::exception& e) {
      std::cerr << "gt_mpi_gather: " << e.what() << "\n";
      return -1;
    }
    return 0;
  }
  if (produce_histogram) {
    // produce_column_histogram (tools/src/gt_mpi_gather.cc:404-411, :600): cells per 100 columns of [0, 4e9), then the partitions of about equal
    // cell count for 128, 64, ... 2 ranks - the column_partitions a loader JSON should carry for that many ranks.  Counted on the GPU.
    try {
      GenomicsDBBCFGenerator gen(loader_json, json_config, "", 0, 0, rank, (size_t)1u << 20, segment_size, "", true, false, true);
      const uint64_t hb = 0, he = 4000000000ull, bin = 100;
      std::vector<uint64_t> counts;
      gen.engine().column_histogram(hb, he, bin, counts);
      for (uint64_t parts : {128ull, 64ull, 32ull, 16ull, 8ull, 4ull, 2ull}) {
        const int64_t n = gdbamd_equi_partition_text(counts.data(), counts.size(), hb, bin, parts, nullptr, 0);
        if (n < 0) { std::cerr << "Requested #equi bins is smaller than allocated bin counts vector, returning\n"; continue; }
        std::string text((size_t)n, '\0');
        gdbamd_equi_partition_text(counts.data(), counts.size(), hb, bin, parts, &text[0], (uint64_t)n);
        std::cout << text;
      }
    } catch (const std::exception& e) {
      std::cerr << "gt_mpi_gather: " << e.what() << "\n";
      return -1;
    }
    return 0;
  }
  try {
    // output file: "vcf_output_filename" (string, or list indexed by rank) else stdout  (json_config.cc:586-607)
    std::string out_name;
    bool index_output = false;
    {
      mini_json::Value q = mini_json::parse_file(json_config);
      index_output = q.IsObject() && q.HasMember("index_output_VCF") && q["index_output_VCF"].GetBool();
      if (q.IsObject() && q.HasMember("vcf_output_filename")) {
        const mini_json::Value& v = q["vcf_output_filename"];
        if (v.IsString()) out_name = v.GetString();
        else if (v.IsArray() && v.Size() > 0) out_name = v[(unsigned)std::min<size_t>((size_t)rank, v.Size() - 1)].GetString();
      }
      if (output_format.empty() && q.IsObject() && q.HasMember("vcf_output_format") && q["vcf_output_format"].IsString()) output_format = q["vcf_output_format"].GetString();
    }
    FILE* out = out_name.empty() ? stdout : fopen(out_name.c_str(), "wb");
    if (!out) { std::cerr << "cannot open " << out_name << "\n"; return -1; }
    const size_t capacity = page_size ? page_size : (size_t)64u << 20;
    // -p is the reference's combined_vcf_records_buffer_size_limit: here it also sizes the pages the device assembles
    if (page_size) setenv("GDBAMD_DEVICE_PAGE_BYTES", std::to_string(page_size).c_str(), 1);
    const auto t0 = std::chrono::steady_clock::now();
    GenomicsDBBCFGenerator gen(loader_json, json_config, "", 0, 0, rank, capacity, segment_size, output_format.c_str(), false, false, true,
                               true);   // (-O z / b: BGZF like the reference's file-writing VCFAdapter, vcf_adapter.cc:340-372)
    std::vector<uint8_t> buf(std::max<size_t>(capacity, 1u << 20));
    size_t total = 0;
    while (!gen.end()) {
      const size_t n = gen.read_and_advance(buf.data(), 0, buf.size());
      if (n == 0) break;
      if (fwrite(buf.data(), 1, n, out) != n) { std::cerr << "short write\n"; return -1; }
      total += n;
    }
    if (out != stdout) fclose(out); else fflush(stdout);
    if (out != stdout && index_output && (output_format == "z" || output_format == "b")) {   // vcf_adapter.cc:275-295: the index from the finished file
      try { if (output_format == "z") build_tbi_index(out_name); else build_csi_index(out_name); }
      catch (const std::exception& e) { std::cerr << "WARNING: error in creating index for output file " << out_name << ": " << e.what() << "\n"; }
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cerr << "GENOMICSDB_TIMER,Rank," << rank << ",scan_and_produce_Broad_GVCF,Wall-clock time(s)," << secs << ",bytes," << total << "\n";
  } catch (const std::exception& e) {
    std::cerr << "gt_mpi_gather: " << e.what() << "\n";
    return -1;
  }
  return 0;
}
