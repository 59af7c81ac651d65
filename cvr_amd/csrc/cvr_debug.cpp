// cvr_debug.cpp -- the library's diagnostic switches: ONE environment variable, CVR_DEBUG="name[=value],name[=value],...", parsed where a
// call of the C ABI begins (cvr_create, cvr_preprocess, cvr_load_image, cvr_power_iteration: debug_refresh) and looked up by name
// everywhere else.  Nothing here changes what a handle computes for valid inputs: the switches choose between equivalent paths (host /
// device planner, staged / one-submission preprocessing, ...), print traces, or set experiment parameters.  What a user of the library
// sets -- devices, caches, the partition rule -- has variables of its own.
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>

namespace cvr {

namespace {
typedef std::map<std::string, std::string> Knobs;
std::mutex                   g_mu;
std::shared_ptr<const Knobs> g_knobs = std::make_shared<const Knobs>();
thread_local std::shared_ptr<const Knobs> t_last;      // keeps the strings a lookup returned alive on this thread
}  // namespace

void debug_refresh()
{
    auto        k = std::make_shared<Knobs>();
    const char *e = getenv("CVR_DEBUG");
    if (e) {
        const char *p = e;
        while (*p) {
            while (*p == ',' || *p == ' ') p++;
            const char *q = p;
            while (*q && *q != ',') q++;
            std::string item(p, q), name = item, value = "1";
            const size_t eq = item.find('=');
            if (eq != std::string::npos) { name = item.substr(0, eq); value = item.substr(eq + 1); }
            for (char &c : name) if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
            if (!name.empty()) (*k)[name] = value;
            p = q;
        }
    }
    std::lock_guard<std::mutex> lk(g_mu);
    g_knobs = k;
}

// the value of `name` in CVR_DEBUG ("1" when it stands there without one), or nullptr; valid until this thread's next lookup
const char *debug_env(const char *name)
{
    {
        std::lock_guard<std::mutex> lk(g_mu);
        t_last = g_knobs;
    }
    const auto it = t_last->find(name);
    return it == t_last->end() ? nullptr : it->second.c_str();
}

}  // namespace cvr
