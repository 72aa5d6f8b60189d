// accounter_parity_test.go — PARITY DUMP of the reference's own Accounter for anyone with a Go toolchain (the build image of
// libnfagg has none, so this file is NOT compiled or run there). It pins what no reference unit test pins: the
// order-dependent fields of model.AccumulateBase (pkg/model/flow_content.go:45-59: last non-zero eth_protocol / dscp /
// sampling, first non-zero MACs), u32 / u64 wrap-around, and "first record stored whole" (pkg/flow/account.go:95), on
// scrambled seeded streams — the same bytes libnfagg's parity tests fold.
//
// 1. In the libnfagg checkout:   python tests/tools/parity_streams.py write /tmp/nfagg_parity
//    (writes stream_<k>.bin = raw 144-byte flow_record_t, and manifest.txt: one line "<k> <records> <max_entries>" each)
// 2. Drop this file into pkg/flow/ of netobserv-ebpf-agent and run
//        NFAGG_PARITY_DIR=/tmp/nfagg_parity go test ./pkg/flow/ -run TestAccounterParityDump -v
//    Every eviction the reference performs (reason full / closing, in order) is written to evictions_<k>.bin as
//    [u32 count][count x (40-byte BpfFlowId + 104-byte BpfFlowMetrics)], host layout.
// 3. Back in the libnfagg checkout (GPU box): python tests/tools/parity_streams.py compare /tmp/nfagg_parity
//    folds the same streams through libnfagg (nfagg_account) and the oracle and compares the three, eviction by eviction,
//    bit for bit (records sorted by key: Go map order is random).
package flow

import (
	"bufio"
	"bytes"
	"encoding/binary"
	"fmt"
	"os"
	"path/filepath"
	"testing"
	"time"

	"github.com/netobserv/netobserv-ebpf-agent/pkg/ebpf"
	"github.com/netobserv/netobserv-ebpf-agent/pkg/metrics"
	"github.com/netobserv/netobserv-ebpf-agent/pkg/model"
)

func TestAccounterParityDump(t *testing.T) {
	dir := os.Getenv("NFAGG_PARITY_DIR")
	if dir == "" {
		t.Skip("NFAGG_PARITY_DIR not set")
	}
	mf, err := os.Open(filepath.Join(dir, "manifest.txt"))
	if err != nil {
		t.Fatal(err)
	}
	defer mf.Close()
	sc := bufio.NewScanner(mf)
	for sc.Scan() {
		var k, n, maxEntries int
		if _, err := fmt.Sscanf(sc.Text(), "%d %d %d", &k, &n, &maxEntries); err != nil {
			continue
		}
		raw, err := os.ReadFile(filepath.Join(dir, fmt.Sprintf("stream_%d.bin", k)))
		if err != nil || len(raw) != n*144 {
			t.Fatalf("stream %d: %v (%d bytes)", k, err, len(raw))
		}
		records := make([]model.RawRecord, n)
		if err := binary.Read(bytes.NewReader(raw), binary.NativeEndian, records); err != nil { // as model.ReadFrom decodes one
			t.Fatal(err)
		}
		now := time.Unix(1_700_000_000, 0)
		acc := NewAccounter(maxEntries, time.Hour, func() time.Time { return now },
			func() time.Duration { return 3_000_000 }, metrics.NoOp(), nil, false)
		inputs := make(chan *model.RawRecord, 50)
		evictor := make(chan []*model.Record, 1024)
		done := make(chan struct{})
		var out bytes.Buffer
		go func() { // every eviction, in order
			for ev := range evictor {
				_ = binary.Write(&out, binary.LittleEndian, uint32(len(ev)))
				for _, r := range ev {
					_ = binary.Write(&out, binary.NativeEndian, r.ID)
					_ = binary.Write(&out, binary.NativeEndian, *r.Metrics.BpfFlowMetrics)
				}
			}
			close(done)
		}()
		go func() {
			acc.Account(inputs, evictor)
			close(evictor)
		}()
		for j := range records {
			inputs <- &records[j]
		}
		close(inputs) // the closing eviction (account.go:73-80)
		<-done
		if err := os.WriteFile(filepath.Join(dir, fmt.Sprintf("evictions_%d.bin", k)), out.Bytes(), 0o644); err != nil {
			t.Fatal(err)
		}
		t.Logf("stream %d: %d records, max_entries %d -> %d bytes of evictions", k, n, maxEntries, out.Len())
	}
	_ = ebpf.BpfFlowId{}
}
