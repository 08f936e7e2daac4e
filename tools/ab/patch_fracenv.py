import sys
s = open(sys.argv[1]).read()
old = "        std::vector<double> frac = {0.02, 0.058, 0.115, 0.19, 0.285, 0.40, 0.535, 0.69, 0.86};\n"
new = old + '''        if (const char *e = getenv("SRS_COMMIT_FRAC")) {     // A/B build only
            frac.clear();
            for (const char *p = e; *p;) { char *q; double v = strtod(p, &q); if (q == p) break; frac.push_back(v); p = (*q == ',') ? q + 1 : q; }
            density = 0.0;
        }
'''
assert old in s
open(sys.argv[2], "w").write(s.replace(old, new))
