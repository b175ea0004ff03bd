// main.cpp -- command-line front end with the reference tool's surface
// (/root/reference/src/main.cpp:101-115 usage, :484-603 flags/validation, :605-659 path expansion, :661-693 model
// dir, :748-775 tile policy, :117-177/:190-416/:793-867 load -> proc -> save pipeline), re-implemented on
// std::thread + the C-ABI engine.  SURVEY.md 8(f-1).  Differences, all deliberate:
//   * -g ids are HIP devices; "-g -1" (ncnn CPU path) is refused: this build has no CPU fallback;
//   * the model is parsed + packed ONCE and reaches the GPUs of "-g 0,1,.." by one RCCL broadcast (rsr_create_group; the
//     reference re-reads x4.bin per GPU, main.cpp:784-786);
//   * a single input image with several GPUs is split by tile rows over all of them (rsr_process_group); directories are
//     dealt image by image from the shared queue exactly like the reference (main.cpp:811-828);
//   * codecs: stb_image / stb_image_write as in the reference (jpg, png, ... in; png, jpg out) + binary pnm; webp in / lossless
//     webp out through the system's libwebp, bound at run time (webp_dl.h) -- without it webp files fail like a broken image;
//   * "-j l:p:s" accepts a single proc count for several GPUs (the reference insists on one per GPU).
#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <queue>
#include <string>
#include <thread>
#include <vector>

#include "image_io.h"
#include "realsr.h"

static void print_usage()
{
    fprintf(stderr, "Usage: realsr-hip -i infile -o outfile [options]...\n\n");
    fprintf(stderr, "  -h                   show this help\n");
    fprintf(stderr, "  -v                   verbose output\n");
    fprintf(stderr, "  -i input-path        input image path (jpg/png/pnm) or directory\n");
    fprintf(stderr, "  -o output-path       output image path (jpg/png) or directory\n");
    fprintf(stderr, "  -s scale             upscale ratio (4, default=4)\n");
    fprintf(stderr, "  -t tile-size         tile size (>=32/0=auto, default=0) can be 0,0,0 for multi-gpu\n");
    fprintf(stderr, "  -m model-path        realsr model path (default=models-DF2K_JPEG)\n");
    fprintf(stderr, "  -g gpu-id            gpu device to use (default=0) can be 0,1,2 for multi-gpu\n");
    fprintf(stderr, "  -j load:proc:save    thread count for load/proc/save (default=1:2:2) can be 1:2,2,2:2 for multi-gpu\n");
    fprintf(stderr, "  -x                   enable tta mode\n");
    fprintf(stderr, "  -f format            output image format (jpg/png/webp, default=ext/png)\n");
}

static std::vector<int> parse_int_list(const char* s)
{
    std::vector<int> v;
    while (s && *s)
    {
        char* e = nullptr;
        const long n = std::strtol(s, &e, 10);
        if (e == s) break;
        v.push_back(int(n));
        s = (*e == ',') ? e + 1 : e;
        if (*e != ',') break;
    }
    return v;
}

static bool is_dir(const std::string& p)
{
    struct stat st;
    return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}
static bool exists(const std::string& p)
{
    struct stat st;
    return stat(p.c_str(), &st) == 0;
}
static int list_files(const std::string& dir, std::vector<std::string>& names)
{
    DIR* d = opendir(dir.c_str());
    if (!d)
    {
        fprintf(stderr, "opendir failed %s\n", dir.c_str());
        return -1;
    }
    while (dirent* e = readdir(d))
    {
        if (e->d_type == DT_REG || (e->d_type == DT_UNKNOWN && !is_dir(dir + "/" + e->d_name))) names.push_back(e->d_name);
    }
    closedir(d);
    std::sort(names.begin(), names.end());
    return 0;
}
static std::string strip_ext(const std::string& n)
{
    const size_t dot = n.find_last_of('.');
    return dot == std::string::npos ? n : n.substr(0, dot);
}
static std::string exe_dir()
{
    char buf[4096];
    const ssize_t n = readlink("/proc/self/exe", buf, sizeof buf - 1);
    if (n <= 0) return ".";
    buf[n] = 0;
    std::string p(buf);
    const size_t s = p.find_last_of('/');
    return s == std::string::npos ? "." : p.substr(0, s);
}
// relative model paths fall back to the executable's directory (filesystem_utils.h:167-173)
static std::string sanitize_filepath(const std::string& p)
{
    if (p.empty() || p[0] == '/' || exists(p)) return p;
    return exe_dir() + "/" + p;
}

struct Task
{
    int id = 0;
    std::string inpath, outpath;
    Image inimage, outimage;
};

class TaskQueue // bounded at 8 like the reference (main.cpp:141)
{
public:
    void put(std::unique_ptr<Task> t)
    {
        std::unique_lock<std::mutex> lk(mu);
        not_full.wait(lk, [&] { return q.size() < 8; });
        q.push(std::move(t));
        not_empty.notify_one();
    }
    std::unique_ptr<Task> get()
    {
        std::unique_lock<std::mutex> lk(mu);
        not_empty.wait(lk, [&] { return !q.empty(); });
        std::unique_ptr<Task> t = std::move(q.front());
        q.pop();
        not_full.notify_one();
        return t;
    }

private:
    std::mutex mu;
    std::condition_variable not_full, not_empty;
    std::queue<std::unique_ptr<Task>> q;
};

static const int kEnd = -233; // same sentinel id as main.cpp:322,350

int main(int argc, char** argv)
{
    std::string inputpath, outputpath, model = "models-DF2K_JPEG", format = "png";
    int scale = 4, jobs_load = 1, jobs_save = 2, verbose = 0, tta_mode = 0;
    std::vector<int> tilesize, gpuid, jobs_proc;

    int opt;
    while ((opt = getopt(argc, argv, "i:o:s:t:m:g:j:f:vxh")) != -1)
    {
        switch (opt)
        {
        case 'i': inputpath = optarg; break;
        case 'o': outputpath = optarg; break;
        case 's': scale = atoi(optarg); break;
        case 't': tilesize = parse_int_list(optarg); break;
        case 'm': model = optarg; break;
        case 'g': gpuid = parse_int_list(optarg); break;
        case 'j':
        {
            const char* c1 = strchr(optarg, ':');
            const char* c2 = c1 ? strchr(c1 + 1, ':') : nullptr;
            if (!c1 || !c2)
            {
                fprintf(stderr, "invalid thread count argument\n");
                return -1;
            }
            jobs_load = atoi(optarg);
            jobs_proc = parse_int_list(c1 + 1);
            jobs_save = atoi(c2 + 1);
            break;
        }
        case 'f': format = optarg; break;
        case 'v': verbose = 1; break;
        case 'x': tta_mode = 1; break;
        case 'h':
        default: print_usage(); return -1;
        }
    }
    if (inputpath.empty() || outputpath.empty())
    {
        print_usage();
        return -1;
    }
    if (scale != 4)
    {
        fprintf(stderr, "invalid scale argument\n");
        return -1;
    }
    if (gpuid.empty()) gpuid.push_back(0);
    const int ngpu = int(gpuid.size());
    for (int g : gpuid)
        if (g < 0)
        {
            fprintf(stderr, "invalid gpu device %d (the ncnn cpu path '-g -1' does not exist in this build: no CPU fallback)\n", g);
            return -1;
        }
    if (!tilesize.empty() && int(tilesize.size()) != ngpu)
    {
        fprintf(stderr, "invalid tilesize argument\n");
        return -1;
    }
    for (int t : tilesize)
        if (t != 0 && t < 32)
        {
            fprintf(stderr, "invalid tilesize argument\n");
            return -1;
        }
    if (jobs_load < 1 || jobs_save < 1)
    {
        fprintf(stderr, "invalid thread count argument\n");
        return -1;
    }
    if (jobs_proc.size() == 1 && ngpu > 1) jobs_proc.assign(size_t(ngpu), jobs_proc[0]);
    if (!jobs_proc.empty() && int(jobs_proc.size()) != ngpu)
    {
        fprintf(stderr, "invalid jobs_proc thread count argument\n");
        return -1;
    }
    for (int j : jobs_proc)
        if (j < 1)
        {
            fprintf(stderr, "invalid jobs_proc thread count argument\n");
            return -1;
        }
    if (!is_dir(outputpath))
    {
        const std::string ext = imgio::lower_ext(outputpath); // format guessed from the output path, whatever -f says
        if (ext == "png") format = "png";
        else if (ext == "webp") format = "webp";
        else if (ext == "jpg" || ext == "jpeg") format = "jpg";
        else
        {
            fprintf(stderr, "invalid outputpath extension type\n");
            return -1;
        }
    }
    if (format != "png" && format != "webp" && format != "jpg")
    {
        fprintf(stderr, "invalid format argument\n");
        return -1;
    }
    if (format == "webp" && !webpdl::api().error.empty())
    {
        fprintf(stderr, "%s\n", webpdl::api().error.c_str());
        return -1;
    }

    // collect input and output file paths
    std::vector<std::string> input_files, output_files;
    if (is_dir(inputpath) && is_dir(outputpath))
    {
        std::vector<std::string> names;
        if (list_files(inputpath, names) != 0) return -1;
        std::string last, last_noext;
        for (const std::string& fn : names)
        {
            const std::string noext = strip_ext(fn);
            std::string outname = noext + "." + format;
            if (noext == last_noext) // sorted list: two inputs would write the same output (main.cpp:625-637)
            {
                const std::string out2 = fn + "." + format;
                fprintf(stderr, "both %s and %s output %s ! %s will output %s\n", fn.c_str(), last.c_str(), outname.c_str(), fn.c_str(), out2.c_str());
                outname = out2;
            }
            else
            {
                last = fn;
                last_noext = noext;
            }
            input_files.push_back(inputpath + "/" + fn);
            output_files.push_back(outputpath + "/" + outname);
        }
    }
    else if (!is_dir(inputpath) && !is_dir(outputpath))
    {
        input_files.push_back(inputpath);
        output_files.push_back(outputpath);
    }
    else
    {
        fprintf(stderr, "inputpath and outputpath must be either file or directory at the same time\n");
        return -1;
    }

    int prepadding = 0;
    if (model.find("models-DF2K") != std::string::npos) prepadding = 10; // also matches models-DF2K_JPEG
    else
    {
        fprintf(stderr, "unknown model dir type\n");
        return -1;
    }
    const std::string parampath = sanitize_filepath(model + "/x4.param");
    const std::string modelpath = sanitize_filepath(model + "/x4.bin");

    if (jobs_proc.empty()) jobs_proc.assign(size_t(ngpu), 2);
    if (tilesize.empty()) tilesize.assign(size_t(ngpu), 0);
    for (size_t i = 0; i < tilesize.size(); i++)
    {
        if (tilesize[i] != 0) continue;
        // the reference's policy on the device's free memory in MB (main.cpp:761-774); an idle MI355X always lands on 200
        long long budget = 0;
        if (rsr_device_memory(gpuid[i], &budget, nullptr) != RSR_OK)
        {
            fprintf(stderr, "%s\n", rsr_last_error(nullptr));
            return -1;
        }
        tilesize[i] = budget > 1900 ? 200 : budget > 550 ? 100 : budget > 190 ? 64 : 32;
    }

    // one context per GPU: parsed + packed once, one RCCL broadcast (rsr_create_group)
    std::vector<rsr_ctx*> ctxs(size_t(ngpu), nullptr);
    if (rsr_create_group(ctxs.data(), gpuid.data(), ngpu, tta_mode, parampath.c_str(), modelpath.c_str()) != RSR_OK)
    {
        fprintf(stderr, "model load failed: %s\n", rsr_last_error(nullptr));
        return -1;
    }
    if (verbose && ngpu > 1) fprintf(stderr, "weights reached %d gpus by: %s\n", ngpu, rsr_group_transport());
    std::vector<std::unique_ptr<RealSR>> realsr;
    for (int i = 0; i < ngpu; i++)
    {
        std::unique_ptr<RealSR> r(new RealSR(ctxs[size_t(i)]));
        r->scale = scale;
        r->tilesize = tilesize[size_t(i)];
        r->prepadding = prepadding;
        // the reference prints one line per tile, "%.2f%%" of (yi * xtiles + xi) / (ytiles * xtiles) (realsr.cpp:481): the same lines
        // here, one per tile of every batch (a batch of tiles runs at once, so they arrive in bursts)
        rsr_set_progress_callback(ctxs[size_t(i)], [](int done, int total, void*) { fprintf(stderr, "%.2f%%\n", total ? 100.f * float(done - 1) / float(total) : 0.f); }, nullptr);
        // every proc thread of this GPU may have a call in flight (small images of concurrent calls are merged into one tile batch: the
        // reference's "-j 4:4:4 for many small images", README.md:61 -- here the more proc threads, the fuller the launches)
        rsr_set_option(ctxs[size_t(i)], "max_lanes", std::max(16, std::min(64, jobs_proc[size_t(i)])));
        // no flag of the reference's surface is taken for it: RSR_PRECISE=1 keeps a byte of rounding residue per element of the residual
        // trunk (rsr_set_option "precise"; what tools/check_real_model.py recommends for weights with a wide output swing)
        if (const char* pe = getenv("RSR_PRECISE"))
            if (pe[0] && pe[0] != '0')
            {
                rsr_set_option(ctxs[size_t(i)], "precise", 1);
                if (verbose) fprintf(stderr, "gpu %d: precise residual trunk (RSR_PRECISE)\n", gpuid[size_t(i)]);
            }
        realsr.push_back(std::move(r));
    }
    // one image, several GPUs: every GPU takes a share of the tile rows
    const bool split_one_image = ngpu > 1 && input_files.size() == 1;

    // load -> proc -> save
    TaskQueue toproc, tosave;
    int failures = 0;
    std::mutex fail_mu;
    const int total_proc = [&] { int n = 0; for (int j : jobs_proc) n += j; return n; }();

    std::vector<std::thread> loaders;
    std::mutex next_mu;
    size_t next = 0;
    for (int t = 0; t < jobs_load; t++)
        loaders.emplace_back([&] {
            for (;;)
            {
                size_t i;
                {
                    std::lock_guard<std::mutex> lk(next_mu);
                    if (next >= input_files.size()) return;
                    i = next++;
                }
                std::unique_ptr<Task> v(new Task);
                v->id = int(i);
                v->inpath = input_files[i];
                v->outpath = output_files[i];
                const std::string err = imgio::load_image(v->inpath, v->inimage);
                if (!err.empty())
                {
                    fprintf(stderr, "decode image %s failed: %s\n", v->inpath.c_str(), err.c_str());
                    std::lock_guard<std::mutex> lk(fail_mu);
                    failures++;
                    continue;
                }
                // an RGBA image cannot be a jpg: write <name>.png instead (main.cpp:278-288)
                const std::string oext = imgio::lower_ext(v->outpath);
                if (v->inimage.elempack == 4 && (oext == "jpg" || oext == "jpeg"))
                {
                    const std::string out2 = v->outpath + ".png";
                    fprintf(stderr, "image %s has alpha channel ! %s will output %s\n", v->inpath.c_str(), v->inpath.c_str(), out2.c_str());
                    v->outpath = out2;
                }
                v->outimage.create(v->inimage.w * scale, v->inimage.h * scale, v->inimage.elempack, true);
                toproc.put(std::move(v));
            }
        });

    std::vector<std::thread> procs;
    for (int g = 0; g < ngpu; g++)
        for (int j = 0; j < jobs_proc[size_t(g)]; j++)
            procs.emplace_back([&, g] {
                for (;;)
                {
                    std::unique_ptr<Task> v = toproc.get();
                    if (v->id == kEnd) return;
                    int prc;
                    if (split_one_image)
                    {
                        std::vector<RealSR*> grp;
                        for (auto& r : realsr) grp.push_back(r.get());
                        prc = RealSR::process_group(grp, v->inimage, v->outimage);
                    }
                    else
                        prc = realsr[size_t(g)]->process(v->inimage, v->outimage);
                    if (prc != RSR_OK)
                    {
                        std::lock_guard<std::mutex> lk(fail_mu);
                        failures++;
                        continue;
                    }
                    tosave.put(std::move(v));
                }
            });

    std::vector<std::thread> savers;
    for (int t = 0; t < jobs_save; t++)
        savers.emplace_back([&] {
            for (;;)
            {
                std::unique_ptr<Task> v = tosave.get();
                if (v->id == kEnd) return;
                const std::string err = imgio::save_image(v->outpath, v->outimage);
                if (!err.empty())
                {
                    fprintf(stderr, "encode image %s failed: %s\n", v->outpath.c_str(), err.c_str());
                    std::lock_guard<std::mutex> lk(fail_mu);
                    failures++;
                }
                else if (verbose)
                    fprintf(stderr, "%s -> %s done\n", v->inpath.c_str(), v->outpath.c_str());
            }
        });

    for (auto& t : loaders) t.join();
    for (int i = 0; i < total_proc; i++)
    {
        std::unique_ptr<Task> e(new Task);
        e->id = kEnd;
        toproc.put(std::move(e));
    }
    for (auto& t : procs) t.join();
    for (int i = 0; i < jobs_save; i++)
    {
        std::unique_ptr<Task> e(new Task);
        e->id = kEnd;
        tosave.put(std::move(e));
    }
    for (auto& t : savers) t.join();
    return failures ? 1 : 0;
}
