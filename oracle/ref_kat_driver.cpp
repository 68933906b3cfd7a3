// ref_kat_driver.cpp -- TEST INFRASTRUCTURE, runs only in the build container.
//
// Known-answer-test driver that is appended (by oracle/Makefile, through a
// pipe) BEHIND the text of the reference program newkmer_10nx.cpp, so that it
// can call the reference's own integerHash / getHash / add_kmer / msca /
// process_qual / process_read and print what they return.  The Makefile's
// stream edit renames the reference `main`, shrinks MAXHASH and routes the
// process_read call inside process_qual through kat_hook() below.  Nothing of
// the reference is stored in this repository; only the numbers this driver
// prints are committed (tests/golden/, via oracle/make_golden.py).
//
// usage: kat_10nx <workdir>
//   in : tree.txt, probes.txt.gz, fmix_in.txt, lookup_in.txt, msca_in.txt,
//        qual_in.txt (seq<TAB>qual per line), reads_in.txt (one sequence per line)
//   out: fmix_out.txt, lookup_out.txt, msca_out.txt, msca_all.txt,
//        qual_out.txt, reads_out.txt
#undef process_read

static int kat_last_start = -1, kat_last_stop = -1, kat_last_final = -1, kat_called = 0;

int kat_hook(string &sequence, string acc, int start, int stop)
{
    kat_called = 1;
    kat_last_start = start;
    kat_last_stop = stop;
    kat_last_final = process_read(sequence, acc, start, stop);
    return kat_last_final;
}

static unsigned long long kat_mix(unsigned long long k)
{
    // independent of the reference: splitmix64 finaliser, used only to weight the checksum
    k ^= k >> 30; k *= 0xbf58476d1ce4e5b9ULL;
    k ^= k >> 27; k *= 0x94d049bb133111ebULL;
    k ^= k >> 31;
    return k;
}

int main(int argc, char **argv)
{
    if (argc < 2) { cerr << "usage: kat <workdir>" << endl; return 2; }
    string wd = argv[1];
    if (wd.back() != '/') wd += "/";

    taxonomy = new Tree1();
    ht = new Hashtable();
    memset(gcount, 0, sizeof(gcount));
    memset(ucount, 0, sizeof(ucount));

    { // tree
        ifstream fin(wd + "tree.txt");
        int a, b;
        while (fin >> a >> b) taxonomy->add_edge(a, b);
    }
    tct = 0;
    process_kmergz(gzopen((wd + "probes.txt.gz").c_str(), "rb"));
    cout << tct << " kmers loaded" << endl;

    { // fmix64
        ifstream fin(wd + "fmix_in.txt");
        ofstream out(wd + "fmix_out.txt");
        unsigned long long k;
        while (fin >> k) out << ht->integerHash(k) << "\n";
    }
    { // getHash
        ifstream fin(wd + "lookup_in.txt");
        ofstream out(wd + "lookup_out.txt");
        unsigned long long k;
        otype org; ptype pos; bool fs;
        while (fin >> k) out << ht->getHash(k, org, pos, fs) << "\n";
    }
    { // msca on explicit pairs
        ifstream fin(wd + "msca_in.txt");
        ofstream out(wd + "msca_out.txt");
        int x, y;
        while (fin >> x >> y) out << taxonomy->msca(x, y) << "\n";
    }
    { // msca over all ordered pairs of nodes 1..MAXTAR-1 -> one checksum
        unsigned long long sum = 0;
        for (int x = 1; x < MAXTAR; x++)
            for (int y = 1; y < MAXTAR; y++)
                sum += kat_mix((unsigned long long)x * MAXTAR + y) * (unsigned long long)taxonomy->msca(x, y);
        ofstream out(wd + "msca_all.txt");
        out << MAXTAR << " " << sum << "\n";
    }
    { // process_qual: (called, start, stop, final_targ)
        ifstream fin(wd + "qual_in.txt");
        ofstream out(wd + "qual_out.txt");
        string line;
        while (getline(fin, line)) {
            size_t tab = line.find('\t');
            if (tab == string::npos) continue;
            string seq = line.substr(0, tab), qual = line.substr(tab + 1);
            kat_called = 0; kat_last_start = kat_last_stop = kat_last_final = -1;
            process_qual("@kat", seq, qual);
            out << kat_called << " " << kat_last_start << " " << kat_last_stop << " " << kat_last_final << "\n";
        }
    }
    { // process_read over whole sequences: final_targ per read, then gcount/ucount
        memset(gcount, 0, sizeof(gcount));
        memset(ucount, 0, sizeof(ucount));
        kmer_seen.clear();
        ifstream fin(wd + "reads_in.txt");
        ofstream out(wd + "reads_out.txt");
        string seq;
        while (getline(fin, seq)) {
            if (seq.empty()) continue;
            out << process_read(seq, "@kat", 0, (int)seq.length() - 1) << "\n";
        }
        ofstream out2(wd + "reads_counts.txt");
        for (int i = 0; i < MAXTAR; i++)
            if (gcount[i] || ucount[i]) out2 << i << " " << gcount[i] << " " << ucount[i] << "\n";
    }
    return 0;
}
