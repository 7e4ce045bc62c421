// ec_hip_host.cpp -- the host-pointer entry points of the HIP backend: how caller memory (ordinary or pinned) gets
// to the kernels and back.  Pageable buffers go through the staging slots' pinned pieces (run_pipeline); pinned
// caller memory is read and written in place by pointer-table kernels (zero copy), optionally mirrored into HBM for
// the checksums of the same trip.  Argument checking happened in ec_api.cpp.  Host code only.
#include "ec_hip.hpp"

#include <algorithm>

namespace gecimpl {

// gec_encode_batch (shard_sums == NULL) / gec_encode_hash_batch
int HipBackend::encode_batch(size_t nblocks, const uint8_t *const *blocks, const size_t *block_len, size_t S, uint8_t *const *parity,
			     uint8_t *shard_sums)
{
	const size_t k = c->k, m = c->m, n = k + m;
	const size_t stripe = n * S;
	ForegroundScope fg(c);
	// the hash kernel's serial chain costs ~3.5 ms per launch whatever the batch, so
	// chunks are 8x larger when checksums are requested
	// caller memory that is pinned end to end needs no host staging, so the only reasons to chunk are the size of
	// the device buffer and the overlap of copy-in / kernels / copy-out between the two slots: 128 MiB chunks
	bool all_pinned = true;
	for (size_t b = 0; b < nblocks && all_pinned; ++b)
		all_pinned = aligned16(blocks[b]) && aligned16(parity[b]) && pinned().contains(blocks[b], block_len[b]) &&
			     pinned().contains(parity[b], m * S);
	if (all_pinned && k <= (size_t)gec::PTR_KMAX) {
		// every buffer is device-addressable: ONE kernel reads the data shards and writes the parity in place
		// over the link; nothing is staged in HBM, no host copy.  With checksums requested the same kernel also
		// lays everything it reads and computes down in HBM (the bytes still cross the link once), chunk by
		// chunk on two streams, and each chunk's shard checksums are computed from there while the next chunk
		// is on the link.
		DeviceGuard dg(c->device);
		if (!dg.ok)
			return fail(GEC_E_DEVICE, "hipSetDevice failed");
		StagingLease lease(c);
		Staging &st = lease.st;
		if (shard_sums && c->sumkind == GEC_SHARDSUM_MLH64) {
			// checksum v3: the link kernel itself leaves the leaf sums of the k shards it reads and the m rows it writes (from
			// the registers that hold them: no mirror in HBM, no second pass), one lane per shard turns them into the 32-byte
			// checksums, which land in the slot's pinned area straight from that kernel.  A PutObject's single block and a
			// coalesced batch of hundreds take the same two launches; chunks only bound the pointer tables.
			const uint32_t nleaf_max = (uint32_t)((S + gec::SHARDSUM_LEAF - 1) / gec::SHARDSUM_LEAF);
			const size_t zch = c->qos_class == GEC_CLASS_BACKGROUND ? chunk_blocks(stripe, nblocks, trip_chunk_bytes(c)) : std::min<size_t>(nblocks, 4096);
			const size_t nz = (nblocks + zch - 1) / zch;
			int rc = st.ensure(nblocks * n * 32 + 64, 0);
			if (!rc)
				rc = st.ensure_tab((nblocks * (k * 12 + m * 8) + 64 * nz) / sizeof(gec::CopyEntry) + 4 * nz + 4);
			if (!rc)
				rc = st.ensure_segments(num_cu);
			if (rc)
				return rc;
			hipStream_t up = st.stream_up ? st.stream_up : st.stream;
			uint8_t *scr = nullptr;
			rc = gecimpl::leaf_scratch(c, up, zch * n * nleaf_max * 8, &scr);
			std::vector<const uint8_t *> in(zch * k);
			std::vector<uint32_t> valid(zch * k);
			std::vector<uint8_t *> out(zch * m);
			for (size_t ci = 0; ci < nz && !rc; ++ci) {
				const size_t b0 = ci * zch, nb = std::min(zch, nblocks - b0);
				background_yield(c);
				for (size_t i = 0; i < nb; ++i) {
					const uint8_t *p = pinned().dev(blocks[b0 + i]);
					uint8_t *q = pinned().dev(parity[b0 + i]);
					const size_t len = block_len[b0 + i];
					for (size_t t = 0; t < k; ++t) {
						in[i * k + t] = p + t * S;
						valid[i * k + t] = (uint32_t)(len > t * S ? std::min(S, len - t * S) : 0);
					}
					for (size_t r = 0; r < m; ++r)
						out[i * m + r] = q + r * S;
				}
				const SumOut so{reinterpret_cast<uint64_t *>(scr), nleaf_max, (uint32_t)n, 0u, true};
				rc = launch_apply_ptrs(c, st, nb, in.data(), valid.data(), out.data(), (int)m, S, c->enc.row(k), up, nullptr, nullptr, 0, nullptr, &so);
				if (!rc)
					rc = mlh_roots_dev(c, nb * n, so.lsum, nleaf_max, nullptr, S, st.h_buf + b0 * n * 32, up);
			}
			const hipError_t e1 = hipStreamSynchronize(up);
			link_release_fire();
			if (rc)
				return rc;
			HIP_TRY(e1);
			std::memcpy(shard_sums, st.h_buf, nblocks * n * 32);
			return GEC_OK;
		}
		if (shard_sums && fused_fits(c, nblocks, S, (int)m, true)) {
			// a PutObject's few blocks: parity AND all k + m checksums from ONE launch (fused.hpp) -- the workgroup that
			// has a tile of the stripe in hand hashes its 14 leaves out of LDS, the block's last workgroup the roots; the
			// checksums land in the slot's pinned area straight from the kernel
			int rc = st.ensure(nblocks * n * 32 + 64, 0);
			if (!rc)
				rc = st.ensure_tab((nblocks * (k * 12 + m * 8) + k * gec::RMAX + 64) / sizeof(gec::CopyEntry) + 8);
			if (!rc)
				rc = st.ensure_segments(num_cu);
			if (rc)
				return rc;
			std::vector<const uint8_t *> in(nblocks * k);
			std::vector<uint32_t> valid(nblocks * k);
			std::vector<uint8_t *> out(nblocks * m);
			for (size_t i = 0; i < nblocks; ++i) {
				const uint8_t *p = pinned().dev(blocks[i]);
				uint8_t *q = pinned().dev(parity[i]);
				for (size_t t = 0; t < k; ++t) {
					in[i * k + t] = p + t * S;
					valid[i * k + t] = (uint32_t)(block_len[i] > t * S ? std::min(S, block_len[i] - t * S) : 0);
				}
				for (size_t r = 0; r < m; ++r)
					out[i * m + r] = q + r * S;
			}
			hipStream_t s1 = st.stream_chain ? st.stream_chain : st.stream;
			rc = launch_fused(c, st, nblocks, in.data(), valid.data(), out.data(), (int)m, S, c->enc.row(k), 1, nullptr, true, st.h_buf, s1);
			const hipError_t e1 = hipStreamSynchronize(s1);
			if (rc)
				return rc;
			HIP_TRY(e1);
			std::memcpy(shard_sums, st.h_buf, nblocks * n * 32);
			return GEC_OK;
		}
		size_t zch = shard_sums || c->qos_class == GEC_CLASS_BACKGROUND ? chunk_blocks(stripe, nblocks, trip_chunk_bytes(c)) : nblocks;
		// with checksums a trip of a few dozen blocks (a batch of the coalescing queue) goes in three chunks rather than one:
		// the checksum kernels of chunk i run beside the link kernel of chunk i+1, and only the last third's are left over
		// when the link falls idle (one chunk: the whole batch's, a fifth of the trip)
		const size_t nz = (nblocks + zch - 1) / zch;
		int rc = st.ensure(shard_sums ? nblocks * n * 32 + 64 : 64, 0);
		if (!rc)
			rc = st.ensure_tab((nblocks * (k * 12 + m * 8) + 64 * nz) / sizeof(gec::CopyEntry) + 4 * nz + 4);
		if (!rc && shard_sums)
			rc = st.ensure_big(2 * (zch * stripe + zch * n * 32));
		if (!rc && shard_sums)
			rc = st.ensure_segments(num_cu);
		if (rc)
			return rc;
		// with checksums: the link kernel on a few CUs of its own, the checksum kernels on the rest (a kernel whose
		// loads share a CU with microsecond-long host reads crawls, see Staging::stream_up); chunk ci+2 reuses the
		// mirror of chunk ci, so its link kernel waits for that chunk's checksums
		hipStream_t up = shard_sums && st.stream_up ? st.stream_up : st.stream;
		hipStream_t chain = shard_sums && st.stream_chain ? st.stream_chain : st.stream2;
		std::vector<const uint8_t *> in(zch * k);
		std::vector<uint32_t> valid(zch * k);
		std::vector<uint8_t *> out(zch * m);
		for (size_t ci = 0; ci < nz && !rc; ++ci) {
			const size_t b0 = ci * zch, nb = std::min(zch, nblocks - b0);
			background_yield(c);
			for (size_t i = 0; i < nb; ++i) {
				const uint8_t *p = pinned().dev(blocks[b0 + i]);
				uint8_t *q = pinned().dev(parity[b0 + i]);
				const size_t len = block_len[b0 + i];
				for (size_t t = 0; t < k; ++t) {
					in[i * k + t] = p + t * S;
					valid[i * k + t] = (uint32_t)(len > t * S ? std::min(S, len - t * S) : 0);
				}
				for (size_t r = 0; r < m; ++r)
					out[i * m + r] = q + r * S;
			}
			uint8_t *mir = shard_sums ? st.d_big + (ci & 1) * (zch * stripe + zch * n * 32) : nullptr;
			if (shard_sums && ci >= 2 && hipStreamWaitEvent(up, st.ev_seg[2 + (ci & 1)], 0) != hipSuccess)
				rc = fail(GEC_E_DEVICE, "hipStreamWaitEvent");
			if (!rc)
				rc = launch_apply_ptrs(c, st, nb, in.data(), valid.data(), out.data(), (int)m, S, c->enc.row(k), up, mir);
			if (rc || !shard_sums)
				continue;
			hipError_t e = hipEventRecord(st.ev_seg[ci & 1], up);
			if (e == hipSuccess)
				e = hipStreamWaitEvent(chain, st.ev_seg[ci & 1], 0);
			if (e != hipSuccess) {
				rc = fail(GEC_E_DEVICE, "chunk event");
				continue;
			}
			// the checksums land in the slot's pinned area straight from the kernel (no copy to launch behind it)
			rc = blake2_dev(c, nb * n, mir, nullptr, nullptr, S, S, st.h_buf + b0 * n * 32, chain, 0, 0, 0, true);
			if (!rc && hipEventRecord(st.ev_seg[2 + (ci & 1)], chain) != hipSuccess)
				rc = fail(GEC_E_DEVICE, "hipEventRecord");
		}
		const hipError_t e1 = hipStreamSynchronize(up);
		link_release_fire();  // the link is free: what is left are the last chunk's checksum kernels
		const hipError_t e2 = shard_sums ? hipStreamSynchronize(chain) : hipSuccess;
		if (rc)
			return rc;
		HIP_TRY(e1);
		HIP_TRY(e2);
		if (shard_sums)
			std::memcpy(shard_sums, st.h_buf, nblocks * n * 32);
		return GEC_OK;
	}
	const size_t ch = chunk_blocks(stripe, nblocks, shard_sums ? trip_chunk_bytes(c) : all_pinned ? pinned_chunk_bytes(c) : kChunkBytes);
	const size_t sums_off = ch * stripe;  // checksum area behind the stripes of a slot
	const size_t nchunks = (nblocks + ch - 1) / ch;
	ForkJoinPool &pool = copy_pool();
	// per chunk: are all its blocks / all its parity buffers in pinned memory the caller registered?
	// Then the DMA engines read / write the caller's memory directly and the staging copy is skipped.
	std::vector<uint8_t> in_pinned(nchunks, 1), out_pinned(nchunks, 1);
	std::vector<size_t> min_len(nchunks, k * S);
	PipeChain chain;
	for (size_t b = 0; b < nblocks; ++b) {
		const size_t ci = b / ch;
		if (in_pinned[ci] && !(aligned16(blocks[b]) && pinned().contains(blocks[b], block_len[b])))
			in_pinned[ci] = 0;
		if (out_pinned[ci] && !(aligned16(parity[b]) && pinned().contains(parity[b], m * S)))
			out_pinned[ci] = 0;
		min_len[ci] = std::min(min_len[ci], block_len[b]);
	}
	return run_pipeline(
		c, nchunks, ch * stripe + (shard_sums ? ch * n * 32 : 0), 0,
		[&](size_t ci, Staging &st) {  // host: user blocks -> pinned, zero-padded to k*S
			(void)st.ensure_tab(2 * ch);  // a failure shows up as "copy table overflow" when the table is used
			if (in_pinned[ci])
				return;
			const size_t b0 = ci * ch, nb = std::min(ch, nblocks - b0);
			pool.parallel_for(nb, [&](size_t i) {
				uint8_t *dst = st.h_buf + i * stripe;
				const size_t len = block_len[b0 + i];
				std::memcpy(dst, blocks[b0 + i], len);
				std::memset(dst + len, 0, k * S - len);
			});
		},
		[&](size_t ci, Staging &st) -> int {  // device: only data shards go H2D, only parity (+sums) comes back
			const size_t b0 = ci * ch, nb = std::min(ch, nblocks - b0);
			if (in_pinned[ci]) {
				// zero padding behind the shortest block of the chunk first (stream order), then
				// every block straight from the caller's memory
				if (min_len[ci] < k * S)
					HIP_TRY(hipMemset2DAsync(st.d_buf + min_len[ci], stripe, 0, k * S - min_len[ci], nb, st.stream));
				// ONE copy_table launch: the kernel reads every block straight from the caller's pinned memory
				std::vector<gec::CopyEntry> ents;
				ents.reserve(nb);
				for (size_t i = 0; i < nb; ++i)
					if (block_len[b0 + i])
						ents.push_back({pinned().dev(blocks[b0 + i]), st.d_buf + i * stripe, block_len[b0 + i]});
				int rct = chain.before(chain.last_in, st.stream);
				if (!rct)
					rct = launch_copy_table(st, ents, st.stream);
				if (!rct)
					rct = chain.after_in(st);
				if (rct)
					return rct;
			} else {
				HIP_TRY(hipMemcpy2DAsync(st.d_buf, stripe, st.h_buf, stripe, k * S, nb, hipMemcpyHostToDevice, st.stream));
			}
			int rc = shard_sums ? encode_hash_dev(c, nb, st.d_buf, stripe, S, st.d_buf + sums_off, st.stream, st)
					    : encode_dev(c, nb, st.d_buf, stripe, S, st.d_buf + k * S, stripe, st.stream);
			if (rc)
				return rc;
			if (out_pinned[ci]) {
				std::vector<gec::CopyEntry> ents;
				ents.reserve(nb);
				for (size_t i = 0; i < nb; ++i)
					ents.push_back({st.d_buf + i * stripe + k * S, pinned().dev(parity[b0 + i]), m * S});
				int rct = chain.before(chain.last_out, st.stream);
				if (!rct)
					rct = launch_copy_table(st, ents, st.stream);
				if (!rct)
					rct = chain.after_out(st);
				if (rct)
					return rct;
			} else {
				HIP_TRY(hipMemcpy2DAsync(st.h_buf + k * S, stripe, st.d_buf + k * S, stripe, m * S, nb, hipMemcpyDeviceToHost, st.stream));
			}
			if (shard_sums)
				HIP_TRY(hipMemcpyAsync(st.h_buf + sums_off, st.d_buf + sums_off, nb * n * 32, hipMemcpyDeviceToHost, st.stream));
			return GEC_OK;
		},
		[&](size_t ci, Staging &st) {  // host: parity (+sums) -> user buffers
			const size_t b0 = ci * ch, nb = std::min(ch, nblocks - b0);
			if (!out_pinned[ci])
				pool.parallel_for(nb, [&](size_t i) { std::memcpy(parity[b0 + i], st.h_buf + i * stripe + k * S, m * S); });
			if (shard_sums)
				std::memcpy(shard_sums + b0 * n * 32, st.h_buf + sums_off, nb * n * 32);
		});
}

// gec_blake2sum_batch (tree == false) / gec_shardsum_batch
int HipBackend::hash_batch(size_t n, const uint8_t *const *msgs, const size_t *lens, uint8_t *out, bool tree)
{
	ForegroundScope fg(c);
	size_t longest = 0;
	bool all_pinned = true;
	for (size_t i = 0; i < n; ++i) {
		longest = std::max(longest, lens[i]);
		if (all_pinned && lens[i] && !(aligned16(msgs[i]) && pinned().contains(msgs[i], lens[i])))
			all_pinned = false;
	}
	// (a) every message in pinned, 16-byte aligned caller memory: NO copy at all -- ONE launch whose lanes
	//     stream their messages straight from host memory over PCIe; only the (offset, length) table and the
	//     32-byte results go through a staging slot.
	// (b) long messages (a BLAKE2b chain costs ~4000 cycles per 128-byte block however many messages run
	//     beside it: 14 ms per MiB): everything is first moved into ONE device buffer through two pinned
	//     staging pieces, then hashed by ONE launch -- chunked launches would pay the chain once per chunk.
	if (all_pinned && tree && n > 1) {
		// Shard checksums of pinned messages: lanes that each stream a 4 KiB leaf out of host memory read the link in
		// 16-byte pieces (22 GiB/s); a copy kernel moves the same bytes coalesced at the link's rate, so the messages
		// go to HBM chunk by chunk (copy_table on the upload stream's CUs) and are hashed there beside the next
		// chunk's transfer.
		DeviceGuard dg(c->device);
		if (!dg.ok)
			return fail(GEC_E_DEVICE, "hipSetDevice failed");
		StagingLease lease(c);
		Staging &st = lease.st;
		const size_t kChunk = trip_chunk_bytes(c);
		const size_t buf_bytes = std::max(kChunk, (longest + 15) / 16 * 16);
		int rc = st.ensure(n * 48 + 64, 0);  // [off][len][out]
		if (!rc)
			rc = st.ensure_tab(n + 8);
		if (!rc)
			rc = st.ensure_big(2 * buf_bytes);
		if (!rc)
			rc = st.ensure_segments(num_cu);
		if (rc)
			return rc;
		hipStream_t up = st.stream_up ? st.stream_up : st.stream;
		hipStream_t chain = st.stream_chain ? st.stream_chain : st.stream2;
		uint64_t *h_off = reinterpret_cast<uint64_t *>(st.h_buf), *h_len = h_off + n;
		uint8_t *h_out = st.h_buf + n * 16;
		size_t ci = 0;
		for (size_t i = 0; i < n && !rc; ++ci) {
			background_yield(c);
			uint8_t *buf = st.d_big + (ci & 1) * buf_bytes;
			std::vector<gec::CopyEntry> ents;
			size_t j = i, bytes = 0, chunk_longest = 0;
			while (j < n && (j == i || bytes + (lens[j] + 15) / 16 * 16 <= buf_bytes)) {
				h_off[j] = bytes;
				h_len[j] = lens[j];
				if (lens[j])
					ents.push_back({pinned().dev(msgs[j]), buf + bytes, lens[j]});
				chunk_longest = std::max(chunk_longest, lens[j]);
				bytes += (lens[j] + 15) / 16 * 16;
				++j;
			}
			if (ci >= 2 && hipStreamWaitEvent(up, st.ev_seg[2 + (ci & 1)], 0) != hipSuccess)
				rc = fail(GEC_E_DEVICE, "hipStreamWaitEvent");
			if (!rc)
				rc = launch_copy_table(st, ents, up);
			hipError_t e = rc ? hipSuccess : hipEventRecord(st.ev_seg[ci & 1], up);
			if (!rc && e == hipSuccess)
				e = hipStreamWaitEvent(chain, st.ev_seg[ci & 1], 0);
			if (!rc && e != hipSuccess)
				rc = fail(GEC_E_DEVICE, "chunk event");
			if (!rc)
				rc = blake2_dev(c, j - i, buf, h_off + i, h_len + i, 0, 0, h_out + 32 * i, chain, 0, 0, 0, true, chunk_longest);
			if (!rc && hipEventRecord(st.ev_seg[2 + (ci & 1)], chain) != hipSuccess)
				rc = fail(GEC_E_DEVICE, "hipEventRecord");
			i = j;
		}
		const hipError_t e1 = hipStreamSynchronize(up), e2 = hipStreamSynchronize(chain);
		if (rc)
			return rc;
		HIP_TRY(e1);
		HIP_TRY(e2);
		std::memcpy(out, h_out, n * 32);
		return GEC_OK;
	}
	if (all_pinned || longest >= (256u << 10) || tree) {
		DeviceGuard dg(c->device);
		if (!dg.ok)
			return fail(GEC_E_DEVICE, "hipSetDevice failed");
		constexpr size_t kPiece = 32ull << 20;
		std::vector<uint64_t> off(n);
		size_t dev_bytes = 0;
		for (size_t i = 0; i < n; ++i) {
			off[i] = dev_bytes;
			dev_bytes += (lens[i] + 15) / 16 * 16;
		}
		if (!all_pinned && dev_bytes > (8ull << 30)) {
			// more than a device buffer should hold at once: halves (each still one launch)
			const size_t h = n / 2;
			int rc = hash_batch(h, msgs, lens, out, tree);
			return rc ? rc : hash_batch(n - h, msgs + h, lens + h, out + 32 * h, tree);
		}
		StagingLease l0(c);
		const size_t meta = n * 16, res = n * 32;
		const size_t meta_off = all_pinned ? 0 : 2 * kPiece;  // h_buf of slot 0: [piece A][piece B][off][len][out]
		int rc = l0.st.ensure(meta_off + meta + res + 64, 0);
		if (rc)
			return rc;
		Staging &st = l0.st;
		uint8_t *d_msgs = nullptr;
		if (!all_pinned)
			HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_msgs), std::max<size_t>(dev_bytes, 16)));
		uint64_t *h_off = reinterpret_cast<uint64_t *>(st.h_buf + meta_off);
		uint64_t *h_len = h_off + n;
		uint8_t *h_out = st.h_buf + meta_off + meta;
		for (size_t i = 0; i < n; ++i) {
			h_off[i] = all_pinned ? reinterpret_cast<uint64_t>(pinned().dev(msgs[i])) : off[i];
			h_len[i] = lens[i];
		}
		auto cleanup = [&](int code) {
			if (d_msgs) {
				(void)hipStreamSynchronize(st.stream);
				(void)hipFree(d_msgs);
			}
			return code;
		};
		if (!all_pinned) {
			// two staging pieces, filled by the copy pool while the other one is on the bus
			ForkJoinPool &pool = copy_pool();
			hipEvent_t done[2] = {st.ev_fork, st.ev_join};
			bool used[2] = {false, false};
			size_t piece = 0;
			for (size_t i = 0; i < n;) {
				// messages (or parts of a long one) that fit the piece
				uint8_t *hp = st.h_buf + (piece & 1) * kPiece;
				if (used[piece & 1]) {
					hipError_t e = hipEventSynchronize(done[piece & 1]);
					if (e != hipSuccess)
						return cleanup(fail(GEC_E_DEVICE, std::string("hipEventSynchronize: ") + hipGetErrorString(e)));
				}
				const size_t d0 = off[i];
				size_t j = i, bytes = 0;
				while (j < n && bytes + (lens[j] + 15) / 16 * 16 <= kPiece) {
					bytes += (lens[j] + 15) / 16 * 16;
					++j;
				}
				if (j == i) {  // one message longer than a piece: stream it through in piece-sized parts
					for (size_t o = 0; o < lens[i]; o += kPiece) {
						hp = st.h_buf + (piece & 1) * kPiece;
						if (used[piece & 1] && hipEventSynchronize(done[piece & 1]) != hipSuccess)
							return cleanup(fail(GEC_E_DEVICE, "hipEventSynchronize failed"));
						const size_t nbytes = std::min(kPiece, lens[i] - o);
						const size_t parts = (nbytes + (1 << 20) - 1) >> 20;
						pool.parallel_for(parts, [&](size_t q) {
							const size_t a = q << 20, b = std::min(nbytes, a + (1 << 20));
							std::memcpy(hp + a, msgs[i] + o + a, b - a);
						});
						hipError_t e = hipMemcpyAsync(d_msgs + off[i] + o, hp, nbytes, hipMemcpyHostToDevice, st.stream);
						if (e == hipSuccess)
							e = hipEventRecord(done[piece & 1], st.stream);
						if (e != hipSuccess)
							return cleanup(fail(GEC_E_DEVICE, std::string("H2D: ") + hipGetErrorString(e)));
						used[piece & 1] = true;
						++piece;
					}
					++i;
					continue;
				}
				pool.parallel_for(j - i, [&](size_t q) { std::memcpy(hp + (off[i + q] - d0), msgs[i + q], lens[i + q]); });
				hipError_t e = hipMemcpyAsync(d_msgs + d0, hp, bytes, hipMemcpyHostToDevice, st.stream);
				if (e == hipSuccess)
					e = hipEventRecord(done[piece & 1], st.stream);
				if (e != hipSuccess)
					return cleanup(fail(GEC_E_DEVICE, std::string("H2D: ") + hipGetErrorString(e)));
				used[piece & 1] = true;
				++piece;
				i = j;
			}
		}
		// the (offset, length) table and the results live in pinned host memory the kernel reads / writes directly
		rc = blake2_dev(c, n, all_pinned ? nullptr : d_msgs, h_off, h_len, 0, 0, h_out, st.stream, 0, 0, 0, tree, longest);
		if (rc)
			return cleanup(rc);
		hipError_t e = hipStreamSynchronize(st.stream);
		if (e != hipSuccess)
			return cleanup(fail(GEC_E_DEVICE, std::string("hipStreamSynchronize: ") + hipGetErrorString(e)));
		std::memcpy(out, h_out, res);
		return cleanup(GEC_OK);
	}
	// (c) many short messages in pageable memory: greedy chunks of <= 4*kChunkBytes of (16-byte aligned) message
	//     slots through the two-slot pipeline (hashing of chunk i overlaps the upload of chunk i+1)
	struct Chunk {
		size_t first, count, bytes;
	};
	std::vector<Chunk> chunks;
	std::vector<uint64_t> slot_off(n);
	size_t max_bytes = 0, max_count = 0;
	for (size_t i = 0; i < n;) {
		Chunk ck{i, 0, 0};
		while (i < n && (ck.count == 0 || ck.bytes + lens[i] <= 4 * kChunkBytes)) {
			slot_off[i] = ck.bytes;
			ck.bytes += (lens[i] + 15) / 16 * 16;
			++ck.count;
			++i;
		}
		chunks.push_back(ck);
		max_bytes = std::max(max_bytes, ck.bytes);
		max_count = std::max(max_count, ck.count);
	}
	// slot layout: [messages][off u64 x count][len u64 x count][out 32 x count]
	const size_t meta_off = (max_bytes + 63) / 64 * 64;
	const size_t out_off = meta_off + 16 * max_count;
	ForkJoinPool &pool = copy_pool();
	return run_pipeline(
		c, chunks.size(), out_off + 32 * max_count, 0,
		[&](size_t ci, Staging &st) {
			const Chunk &ck = chunks[ci];
			uint64_t *offs = reinterpret_cast<uint64_t *>(st.h_buf + meta_off);
			uint64_t *ls = offs + ck.count;
			pool.parallel_for(ck.count, [&](size_t i) {
				std::memcpy(st.h_buf + slot_off[ck.first + i], msgs[ck.first + i], lens[ck.first + i]);
				offs[i] = slot_off[ck.first + i];
				ls[i] = lens[ck.first + i];
			});
		},
		[&](size_t ci, Staging &st) -> int {
			const Chunk &ck = chunks[ci];
			HIP_TRY(hipMemcpyAsync(st.d_buf, st.h_buf, ck.bytes, hipMemcpyHostToDevice, st.stream));
			HIP_TRY(hipMemcpyAsync(st.d_buf + meta_off, st.h_buf + meta_off, 16 * ck.count, hipMemcpyHostToDevice, st.stream));
			const uint64_t *d_off = reinterpret_cast<const uint64_t *>(st.d_buf + meta_off);
			int rc = blake2_dev(c, ck.count, st.d_buf, d_off, d_off + ck.count, 0, 0, st.d_buf + out_off, st.stream);
			if (rc)
				return rc;
			HIP_TRY(hipMemcpyAsync(st.h_buf + out_off, st.d_buf + out_off, 32 * ck.count, hipMemcpyDeviceToHost, st.stream));
			return GEC_OK;
		},
		[&](size_t ci, Staging &st) {
			const Chunk &ck = chunks[ci];
			std::memcpy(out + 32 * ck.first, st.h_buf + out_off, 32 * ck.count);
		});
}

int HipBackend::decode_verify_batch(size_t nblocks, const uint8_t *const *shards, size_t S, const size_t *block_len,
				    uint8_t *const *rebuilt, uint8_t *shard_sums, uint8_t *block_sums)
{
	const size_t k = c->k, n = c->k + c->m;
	ForegroundScope fg(c);
	// more than one device buffer should hold: halves
	const size_t kMaxBytes = 6ull << 30;
	if (nblocks > 1 && nblocks * n * S > kMaxBytes) {
		const size_t h = nblocks / 2;
		int rc = decode_verify_batch(h, shards, S, block_len, rebuilt, shard_sums, block_sums);
		if (rc)
			return rc;
		return decode_verify_batch(nblocks - h, shards + h * n, S, block_len ? block_len + h : nullptr,
					   rebuilt ? rebuilt + h * n : nullptr, shard_sums + h * n * 32,
					   block_sums ? block_sums + h * 32 : nullptr);
	}
	// -- per block: which shards are read (the crate's rule: the first k present), which data shards are rebuilt;
	//    blocks are laid out on the device bucket by bucket (one erasure pattern each), a block's stripe holding
	//    its k data slots followed by one slot per parity shard it is decoded from
	struct Bucket {
		std::shared_ptr<const Plan> plan;
		std::vector<size_t> ids;
		size_t base = 0, stripe = 0, npar = 0;
		size_t j0 = 0;  // first missing data slot (k: none)
		std::vector<size_t> rq;  // staging slot of every rebuilt shard that cannot be written to its buffer directly
	};
	std::map<std::string, Bucket> buckets;
	for (size_t b = 0; b < nblocks; ++b) {
		std::string key(n, 0);
		size_t np = 0;
		for (size_t j = 0; j < n; ++j) {
			key[j] = shards[b * n + j] ? 1 : 0;
			np += key[j];
		}
		if (np < k)
			return fail(GEC_E_TOO_FEW_PRESENT, "fewer than k shards present");
		if (block_len && block_len[b] > k * S)
			return fail(GEC_E_INCORRECT_SHARD_SIZE, "block longer than k*S");
		buckets[key].ids.push_back(b);
	}
	size_t dev_bytes = 0, nup = 0, nreb = 0;
	for (auto &kv : buckets) {
		Bucket &bk = kv.second;
		int rc = get_plan(c, reinterpret_cast<const uint8_t *>(kv.first.data()), true, bk.plan);
		if (rc)
			return rc;
		bk.npar = bk.plan->missing.size();  // as many parity inputs as data shards to rebuild
		bk.j0 = bk.npar ? (size_t)bk.plan->missing[0] : k;
		bk.stripe = (k + bk.npar) * S;
		bk.base = dev_bytes;
		dev_bytes += bk.ids.size() * bk.stripe;
		nup += bk.ids.size() * k;
		nreb += bk.ids.size() * bk.npar;
		for (size_t b : bk.ids)
			for (int j : bk.plan->missing)
				if (!rebuilt || !rebuilt[b * n + j])
					return fail(GEC_E_INVALID_ARG, "NULL output for a missing data shard");
	}
	DeviceGuard dg(c->device);
	if (!dg.ok)
		return fail(GEC_E_DEVICE, "hipSetDevice failed");
	bool all_pinned = true;
	for (size_t b = 0; b < nblocks && all_pinned; ++b)
		for (size_t j = 0; j < n && all_pinned; ++j)
			if (shards[b * n + j])
				all_pinned = aligned16(shards[b * n + j]) && pinned().contains(shards[b * n + j], S);
	StagingLease lease(c);
	Staging &st = lease.st;
	// A GetObject's few blocks, every buffer pinned, no end-to-end hash asked for: ONE launch (fused.hpp) reads the k shards
	// of every block out of the caller's memory, returns every shard's checksum, and writes the missing data shards -- each
	// block with the coefficient set of its own erasure pattern -- straight into the buffers they are wanted in.  Nothing
	// is staged in HBM, nothing is copied home, however many patterns the batch has.
	if (all_pinned && !block_sums && buckets.size() <= 0xffff) {
		size_t max_miss = 0;
		bool out_pinned = true;
		for (auto &kv : buckets) {
			max_miss = std::max(max_miss, kv.second.npar);
			for (size_t b : kv.second.ids)
				for (int j : kv.second.plan->missing)
					out_pinned = out_pinned && aligned16(rebuilt[b * n + j]) && pinned().contains(rebuilt[b * n + j], S);
		}
		if (out_pinned && max_miss <= (size_t)gec::RMAX && fused_fits(c, nblocks, S, (int)max_miss, false)) {
			int frc = st.ensure(nblocks * k * 32 + 64, 0);
			if (!frc)
				frc = st.ensure_tab((nblocks * (k * 12 + max_miss * 8 + 2) + buckets.size() * k * gec::RMAX + 64) / sizeof(gec::CopyEntry) + 8);
			if (!frc)
				frc = st.ensure_segments(num_cu);
			if (frc)
				return frc;
			std::vector<const uint8_t *> in(nblocks * k);
			std::vector<uint32_t> valid(nblocks * k, (uint32_t)S);
			std::vector<uint8_t *> out(nblocks * max_miss, nullptr);
			std::vector<uint16_t> pat(nblocks);
			std::vector<uint8_t> sets(buckets.size() * max_miss * k, 0);
			size_t pi = 0, i = 0;  // blocks go bucket by bucket: a workgroup rebuilds its tables only when the pattern changes
			for (auto &kv : buckets) {
				Bucket &bk = kv.second;
				for (size_t r = 0; r < bk.npar; ++r)
					std::memcpy(&sets[(pi * max_miss + r) * k], bk.plan->rows.v.data() + r * k, k);
				for (size_t b : bk.ids) {
					pat[i] = (uint16_t)pi;
					for (size_t t = 0; t < k; ++t)
						in[i * k + t] = pinned().dev(shards[b * n + bk.plan->valid[t]]);
					for (size_t r = 0; r < bk.npar; ++r)
						out[i * max_miss + r] = pinned().dev(rebuilt[b * n + bk.plan->missing[r]]);
					++i;
				}
				++pi;
			}
			hipStream_t s1 = st.stream_chain ? st.stream_chain : st.stream;
			frc = launch_fused(c, st, nblocks, in.data(), valid.data(), out.data(), (int)max_miss, S, sets.data(), buckets.size(), pat.data(),
					   false, st.h_buf, s1);
			const hipError_t e1 = hipStreamSynchronize(s1);
			if (frc)
				return frc;
			HIP_TRY(e1);
			i = 0;
			for (auto &kv : buckets)
				for (size_t b : kv.second.ids) {
					for (size_t t = 0; t < k; ++t)
						std::memcpy(shard_sums + 32 * (b * n + kv.second.plan->valid[t]), st.h_buf + 32 * (i * k + t), 32);
					++i;
				}
			return GEC_OK;
		}
	}
	constexpr size_t kPiece = 32ull << 20;
	// host staging: [piece A][piece B] (pageable shards only) [shard off | shard len | block off | block len][shard sums][block sums][rebuilt]
	const size_t tab_off = all_pinned ? 0 : 2 * kPiece;
	const size_t tab_bytes = (nup * 2 + nblocks * 2) * 8;
	const size_t ssum_off = tab_off + tab_bytes, bsum_off = ssum_off + nup * 32;
	const size_t reb_off = (bsum_off + nblocks * 32 + 63) / 64 * 64;
	int rc = st.ensure(reb_off + nreb * S + 64, 0);
	if (rc)
		return rc;
	const size_t state_off = (dev_bytes + 63) / 64 * 64;  // chaining values of the segmented block checksums
	rc = st.ensure_big(state_off + nblocks * 64 + 64);
	if (rc)
		return rc;
	rc = st.ensure_tab(nup + nreb + (nblocks * (k * 12 + gec::RMAX * 8 + 2) + buckets.size() * k * gec::RMAX + 1024) / sizeof(gec::CopyEntry) + 64);
	if (rc)
		return rc;
	rc = st.ensure_segments(num_cu);
	if (rc)
		return rc;
	uint64_t *h_soff = reinterpret_cast<uint64_t *>(st.h_buf + tab_off), *h_slen = h_soff + nup;
	uint64_t *h_boff = h_slen + nup, *h_blen = h_boff + nblocks;
	// -- block table (the blocks that need no decode first)
	size_t bi = 0, longest = 0;
	std::vector<size_t> block_order(nblocks);
	for (int pass = 0; pass < 2; ++pass) {
		for (auto &kv : buckets) {
			if ((kv.second.npar == 0) != (pass == 0))
				continue;
			for (size_t i = 0; i < kv.second.ids.size(); ++i) {
				h_boff[bi] = kv.second.base + i * kv.second.stripe;
				h_blen[bi] = block_len ? block_len[kv.second.ids[i]] : 0;
				longest = std::max<size_t>(longest, h_blen[bi]);
				block_order[bi++] = kv.second.ids[i];
			}
		}
	}
	// Upload stages.  A block's own checksum is one serial BLAKE2b chain -- ~10.6 ms per MiB whatever runs beside it,
	// as long as the link needs for 512 such blocks -- so the chains must run WHILE the blocks arrive.  Data slot s
	// belongs to stage(s) = the st with k*st/nseg <= s < k*(st+1)/nseg, and after every stage ONE segment launch
	// advances the chain of every block of the batch over the 128-byte blocks of that stage's slots, in lock step.
	// What a stage uploads is decided per block by its first missing data slot j0 (j0 = k for a block that needs no
	// decode):
	//   - data slot s < j0 travels in stage(s): it is there when the chains reach it;
	//   - everything else the block is decoded from -- its data slots beyond j0, its parity inputs -- travels in
	//     stage(j0): when the chains reach slot j0 the block is complete on the device, its decode (one launch per
	//     erasure pattern, on the chain stream, right before the segment) has filled slot j0 and every later
	//     missing slot, and the rebuilt shards start for home on a third stream while later stages still arrive.
	// This is earliest-deadline-first for "slot s must be hashable at chain step s": a batch where EVERY block has
	// lost data shards (a node of each stripe down) finishes a chain step after the last byte arrived, like a
	// healthy one, instead of upload + decode + a whole chain (24.7 -> see profiles/r03_get_degraded.txt).
	// One stage unless every shard is pinned (the staged path uploads dense device ranges block by block) and the
	// chains are long enough to be worth hiding.  (One stage -- upload everything, then hash -- was the A/B: slower from 24 blocks on.)
	const int seg_max = (int)Staging::kMaxSeg;
	const size_t nseg = (all_pinned && block_sums && longest >= (256u << 10)) ? std::min<size_t>(k, (size_t)seg_max) : 1;
	auto stage_of_slot = [&](size_t slot) {
		size_t sg = 0;
		while (sg + 1 < nseg && k * (sg + 1) / nseg <= slot)
			++sg;
		return sg;
	};
	// -- upload list, in device order
	struct Up {
		const uint8_t *src;
		size_t dst;  // byte offset in d_big
		size_t idx;  // (b*n + j): where the shard's checksum goes
		size_t stage;
	};
	std::vector<Up> ups;
	ups.reserve(nup);
	for (auto &kv : buckets) {
		Bucket &bk = kv.second;
		for (size_t i = 0; i < bk.ids.size(); ++i) {
			const size_t b = bk.ids[i];
			size_t q = 0;  // parity input slot
			for (size_t t = 0; t < k; ++t) {
				const int j = bk.plan->valid[t];
				const size_t slot = (size_t)j < k ? (size_t)j : k + q++;
				const uint8_t *p = shards[b * n + j];
				const size_t stage = nseg == 1 ? 0 : stage_of_slot(std::min<size_t>(slot, bk.j0));  // parity inputs: slot >= k > j0
				ups.push_back({p, bk.base + i * bk.stripe + slot * S, b * n + j, stage});
			}
		}
	}
	std::sort(ups.begin(), ups.end(), [](const Up &a, const Up &b) { return a.dst < b.dst; });
	for (size_t i = 0; i < ups.size(); ++i) {
		h_soff[i] = ups[i].dst;
		h_slen[i] = S;
	}
	auto hip_fail = [&](hipError_t e, const char *what) { return fail(GEC_E_DEVICE, std::string(what) + ": " + hipGetErrorString(e)); };
	hipEvent_t ev_up = nullptr, ev_sh = nullptr;
	HIP_TRY(hipEventCreateWithFlags(&ev_up, hipEventDisableTiming));
	HIP_TRY(hipEventCreateWithFlags(&ev_sh, hipEventDisableTiming));
	// staged upload: its own (CU-masked) stream pair when there is one
	hipStream_t up_stream = nseg > 1 && st.stream_up ? st.stream_up : st.stream;
	hipStream_t chain_stream = nseg > 1 && st.stream_chain ? st.stream_chain : st.stream2;
	auto finish = [&](int code) {
		(void)hipStreamSynchronize(up_stream);
		(void)hipStreamSynchronize(chain_stream);
		(void)hipStreamSynchronize(st.stream);
		(void)hipStreamSynchronize(st.stream2);
		(void)hipStreamSynchronize(st.stream3);
		if (st.stream_down)
			(void)hipStreamSynchronize(st.stream_down);
		(void)hipEventDestroy(ev_up);
		(void)hipEventDestroy(ev_sh);
		return code;
	};
	// decode of one erasure pattern, in place in the bucket's stripes; the rebuilt shards' way home is appended to `outs`
	size_t rq = 0;
	auto decode_range = [&](Bucket &bk, size_t first, size_t count, hipStream_t s, std::vector<gec::CopyEntry> &outs) -> int {
		std::vector<size_t> in_off(k), out_off(bk.npar);
		size_t q = 0;
		for (size_t t = 0; t < k; ++t) {
			const int j = bk.plan->valid[t];
			in_off[t] = ((size_t)j < k ? (size_t)j : k + q++) * S;
		}
		for (size_t r = 0; r < bk.npar; ++r)
			out_off[r] = (size_t)bk.plan->missing[r] * S;  // rebuilt in place, in the block's data area
		uint8_t *base = st.d_big + bk.base + first * bk.stripe;
		int drc = launch_apply(c, base, bk.stripe, base, bk.stripe, nullptr, 0, S, count, in_off.data(), out_off.data(), (int)bk.npar,
				       bk.plan->rows.v.data(), gec::MODE_STORE, s);
		if (drc)
			return drc;
		for (size_t i = first; i < first + count; ++i)
			for (size_t r = 0; r < bk.npar; ++r) {
				uint8_t *dst = rebuilt[bk.ids[i] * n + bk.plan->missing[r]];
				const bool direct = aligned16(dst) && pinned().contains(dst, S);
				outs.push_back({st.d_big + bk.base + i * bk.stripe + out_off[r], direct ? pinned().dev(dst) : st.h_buf + reb_off + rq * S, S});
				bk.rq.push_back(rq++);
			}
		return GEC_OK;
	};
	auto decode_bucket = [&](Bucket &bk, hipStream_t s, std::vector<gec::CopyEntry> &outs) -> int {
		return decode_range(bk, 0, bk.ids.size(), s, outs);
	};
	// A big batch without block checksums (the default read path: no end-to-end hash at the requester) has no long chains
	// to hide, and used to do one thing after the other: upload everything (the link's 10.7 ms for 512 blocks), THEN the
	// shard checksums, the decodes, the rebuilt shards' way home -- 17 ms with 4 of 16 nodes down.  In pieces instead:
	// piece c+1 is on its way up while piece c is hashed and decoded and piece c-1's rebuilt shards travel down (the link
	// is full duplex) -- three streams, one event per piece and direction.
	// (24 blocks and four pieces: a piece keeps at least a dozen blocks -- below that the per-piece launches cost more than the
	// overlap returns; more than four pieces did not shorten a 512-block trip, profiles/r04_trip_bench.txt)
	constexpr size_t kGetPiecesMin = 24, kGetPieces = 4;
	if (all_pinned && !block_sums && nblocks >= kGetPiecesMin) {
		struct Part {
			Bucket *bk;
			size_t first, count;
		};
		const size_t npieces = std::max<size_t>(1, std::min<size_t>({(size_t)Staging::kMaxSeg, kGetPieces, nblocks / 12}));
		const size_t per_piece = (nblocks + npieces - 1) / npieces;
		std::vector<std::vector<Part>> pieces(1);
		size_t in_piece = 0;
		for (auto &kv : buckets) {
			Bucket &bk = kv.second;
			for (size_t f = 0; f < bk.ids.size();) {
				if (in_piece == per_piece) {
					pieces.emplace_back();
					in_piece = 0;
				}
				const size_t cnt = std::min(bk.ids.size() - f, per_piece - in_piece);
				pieces.back().push_back({&bk, f, cnt});
				f += cnt;
				in_piece += cnt;
			}
		}
		hipStream_t up_s = st.stream_up ? st.stream_up : st.stream;
		hipStream_t chain_s = st.stream_chain ? st.stream_chain : st.stream2;
		hipStream_t down_s = st.stream_down ? st.stream_down : st.stream3;
		size_t up_i = 0;  // ups[] is sorted by device address = bucket by bucket, block by block: a piece is a run of it
		for (size_t pc = 0; pc < pieces.size(); ++pc) {
			size_t lo = ~(size_t)0, hi = 0;  // device byte range of the piece
			for (const Part &pt : pieces[pc]) {
				lo = std::min(lo, pt.bk->base + pt.first * pt.bk->stripe);
				hi = std::max(hi, pt.bk->base + (pt.first + pt.count) * pt.bk->stripe);
			}
			const size_t i0 = up_i;
			std::vector<gec::CopyEntry> ents;
			while (up_i < ups.size() && ups[up_i].dst < hi) {
				size_t run = 1;
				while (up_i + run < ups.size() && ups[up_i + run].dst < hi && ups[up_i + run].src == ups[up_i].src + run * S &&
				       ups[up_i + run].dst == ups[up_i].dst + run * S)
					++run;
				ents.push_back({pinned().dev(ups[up_i].src), st.d_big + ups[up_i].dst, run * S});
				up_i += run;
			}
			(void)lo;
			rc = launch_copy_table(st, ents, up_s);
			if (rc)
				return finish(rc);
			hipError_t es = hipEventRecord(st.ev_seg[pc], up_s);
			if (es == hipSuccess)
				es = hipStreamWaitEvent(chain_s, st.ev_seg[pc], 0);
			if (es != hipSuccess)
				return finish(hip_fail(es, "piece event"));
			// the piece's shard checksums, straight into the slot's pinned area
			rc = blake2_dev(c, up_i - i0, st.d_big, h_soff + i0, h_slen + i0, 0, 0, st.h_buf + ssum_off + 32 * i0, chain_s, 0, 0, 0, true, S);
			if (rc)
				return finish(rc);
			// the piece's decodes: ONE launch whatever mix of erasure patterns its blocks have (a coefficient set per block,
			// the pointer-table kernel over the stripes in HBM, rebuilt in place), instead of one launch per pattern
			std::vector<gec::CopyEntry> outs;
			{
				size_t max_miss = 0, nb_dec = 0;
				for (const Part &pt : pieces[pc])
					if (pt.bk->npar) {
						max_miss = std::max(max_miss, pt.bk->npar);
						nb_dec += pt.count;
					}
				if (nb_dec && max_miss <= (size_t)gec::RMAX && k <= (size_t)gec::PTR_KMAX) {
					std::vector<const uint8_t *> in(nb_dec * k);
					std::vector<uint32_t> valid(nb_dec * k, (uint32_t)S);
					std::vector<uint8_t *> outp(nb_dec * max_miss, nullptr);
					std::vector<uint16_t> pat(nb_dec);
					std::vector<uint8_t> sets;
					size_t bi2 = 0, pi = 0;
					for (const Part &pt : pieces[pc]) {
						Bucket &bk = *pt.bk;
						if (!bk.npar)
							continue;
						sets.resize((pi + 1) * max_miss * k, 0);
						for (size_t r = 0; r < bk.npar; ++r)
							std::memcpy(&sets[(pi * max_miss + r) * k], bk.plan->rows.v.data() + r * k, k);
						for (size_t i = pt.first; i < pt.first + pt.count; ++i, ++bi2) {
							uint8_t *stripe0 = st.d_big + bk.base + i * bk.stripe;
							size_t q = 0;
							for (size_t t = 0; t < k; ++t) {
								const int j = bk.plan->valid[t];
								in[bi2 * k + t] = stripe0 + ((size_t)j < k ? (size_t)j : k + q++) * S;
							}
							pat[bi2] = (uint16_t)pi;
							for (size_t r = 0; r < bk.npar; ++r) {
								uint8_t *slot = stripe0 + (size_t)bk.plan->missing[r] * S;
								outp[bi2 * max_miss + r] = slot;
								uint8_t *dst = rebuilt[bk.ids[i] * n + bk.plan->missing[r]];
								const bool direct = aligned16(dst) && pinned().contains(dst, S);
								outs.push_back({slot, direct ? pinned().dev(dst) : st.h_buf + reb_off + rq * S, S});
								bk.rq.push_back(rq++);
							}
						}
						++pi;
					}
					rc = launch_apply_ptrs(c, st, nb_dec, in.data(), valid.data(), outp.data(), (int)max_miss, S, sets.data(), chain_s, nullptr,
							       nullptr, pi, pat.data());
					if (rc)
						return finish(rc);
				} else {
					for (const Part &pt : pieces[pc])
						if (pt.bk->npar) {
							rc = decode_range(*pt.bk, pt.first, pt.count, chain_s, outs);
							if (rc)
								return finish(rc);
						}
				}
			}
			if (!outs.empty()) {
				es = hipEventRecord(st.ev_dec[pc], chain_s);
				if (es == hipSuccess)
					es = hipStreamWaitEvent(down_s, st.ev_dec[pc], 0);
				if (es != hipSuccess)
					return finish(hip_fail(es, "decode event"));
				rc = launch_copy_table(st, outs, down_s);
				if (rc)
					return finish(rc);
			}
		}
		hipError_t e1 = hipStreamSynchronize(up_s);
		if (nreb == 0)
			link_release_fire();  // nothing goes home: the link is free, the last piece's checksums are what is left
		if (e1 == hipSuccess)
			e1 = hipStreamSynchronize(chain_s);
		if (e1 == hipSuccess)
			e1 = hipStreamSynchronize(down_s);
		if (e1 != hipSuccess)
			return finish(hip_fail(e1, "hipStreamSynchronize"));
		for (size_t i = 0; i < ups.size(); ++i)
			std::memcpy(shard_sums + 32 * ups[i].idx, st.h_buf + ssum_off + 32 * i, 32);
		for (auto &kv : buckets) {
			Bucket &bk = kv.second;
			size_t w = 0;
			for (size_t i = 0; i < bk.ids.size(); ++i)
				for (size_t r = 0; r < bk.npar; ++r, ++w) {
					uint8_t *dst = rebuilt[bk.ids[i] * n + bk.plan->missing[r]];
					if (!(aligned16(dst) && pinned().contains(dst, S)))
						std::memcpy(dst, st.h_buf + reb_off + bk.rq[w] * S, S);
				}
		}
		return finish(GEC_OK);
	}
	const bool staged = all_pinned && nseg > 1;
	hipStream_t down_stream = staged && st.stream_down ? st.stream_down : st.stream;
	if (all_pinned) {
		std::vector<std::vector<gec::CopyEntry>> ents(nseg);
		for (size_t i = 0; i < ups.size();) {  // merge neighbours (the data shards of a block are slices of one buffer)
			size_t run = 1;
			while (i + run < ups.size() && ups[i + run].stage == ups[i].stage && ups[i + run].src == ups[i].src + run * S &&
			       ups[i + run].dst == ups[i].dst + run * S)
				++run;
			ents[ups[i].stage].push_back({pinned().dev(ups[i].src), st.d_big + ups[i].dst, run * S});
			i += run;
		}
		for (size_t sg = 0; sg < nseg; ++sg) {
			rc = launch_copy_table(st, ents[sg], up_stream);
			if (rc)
				return finish(rc);
			if (!staged)
				break;
			hipError_t es = hipEventRecord(st.ev_seg[sg], up_stream);
			if (es == hipSuccess)
				es = hipStreamWaitEvent(chain_stream, st.ev_seg[sg], 0);
			if (es != hipSuccess)
				return finish(hip_fail(es, "stage event"));
			// the blocks whose first missing slot lies in this stage are complete now: decode them, send the rebuilt
			// shards home on the down stream (beside the stages still to come: the link is full duplex)
			std::vector<gec::CopyEntry> outs;
			for (auto &kv : buckets)
				if (kv.second.npar && stage_of_slot(kv.second.j0) == sg) {
					rc = decode_bucket(kv.second, chain_stream, outs);
					if (rc)
						return finish(rc);
				}
			if (!outs.empty()) {
				es = hipEventRecord(st.ev_dec[sg], chain_stream);
				if (es == hipSuccess)
					es = hipStreamWaitEvent(down_stream, st.ev_dec[sg], 0);
				if (es != hipSuccess)
					return finish(hip_fail(es, "decode event"));
				// paced: the copies have a chain segment's time (~1 ms) to finish in, and at the link's full write rate
				// they stop the chains for exactly as long as they take (segments 1.07 -> 1.2-1.7 ms: the fabric's queues
				// towards the link fill up and every load behind them waits) -- at half the rate the chains do not notice
				const unsigned home_wgs = 16, rate = env().home_rate_gbps;
				rc = launch_copy_table(st, outs, down_stream, rate ? home_wgs : 0, rate ? home_wgs * 16384u / rate : 0);
				if (rc)
					return finish(rc);
			}
			// every chain advances over what is on the device now: whole 128-byte blocks below the end of this stage's
			// last slot (the final stage finishes every message)
			const uint64_t blk0 = (k * sg / nseg) * S / 128;
			const uint64_t blk1 = sg + 1 == nseg ? ~0ull : (k * (sg + 1) / nseg) * S / 128;
			rc = blake2_dev(c, nblocks, st.d_big, h_boff, h_blen, 0, 0, st.h_buf + bsum_off, chain_stream, 0, 0, 0, false, 0,
					reinterpret_cast<uint64_t *>(st.d_big + state_off), blk0, blk1);
			if (rc)
				return finish(rc);
		}
	} else {
		// pageable shards: the pieces are images of dense device ranges, filled by the copy pool while the other is on the bus
		ForkJoinPool &pool = copy_pool();
		hipEvent_t done[2] = {st.ev_fork, st.ev_join};
		bool used[2] = {false, false};
		size_t piece = 0;
		for (size_t i = 0; i < ups.size();) {
			uint8_t *hp = st.h_buf + (piece & 1) * kPiece;
			if (used[piece & 1]) {
				hipError_t e = hipEventSynchronize(done[piece & 1]);
				if (e != hipSuccess)
					return finish(hip_fail(e, "hipEventSynchronize"));
			}
			const size_t d0 = ups[i].dst;
			size_t j = i;
			while (j < ups.size() && ups[j].dst + S - d0 <= kPiece)
				++j;
			if (j == i)
				return finish(fail(GEC_E_INVALID_ARG, "shard larger than a staging piece"));
			const size_t bytes = ups[j - 1].dst + S - d0;
			pool.parallel_for(j - i, [&](size_t q) { std::memcpy(hp + (ups[i + q].dst - d0), ups[i + q].src, S); });
			hipError_t e = hipMemcpyAsync(st.d_big + d0, hp, bytes, hipMemcpyHostToDevice, st.stream);
			if (e == hipSuccess)
				e = hipEventRecord(done[piece & 1], st.stream);
			if (e != hipSuccess)
				return finish(hip_fail(e, "H2D"));
			used[piece & 1] = true;
			++piece;
			i = j;
		}
	}
	// -- everything is on the device.  Shard checksums on their own stream (they depend on nothing else); on the
	//    main stream: decode per bucket, then the checksums of the blocks not hashed yet, then the rebuilt shards go home.
	hipError_t e = hipEventRecord(ev_up, up_stream);
	if (e == hipSuccess)
		e = hipStreamWaitEvent(st.stream3, ev_up, 0);
	if (e == hipSuccess && up_stream != st.stream)
		e = hipStreamWaitEvent(st.stream, ev_up, 0);
	if (e != hipSuccess)
		return finish(hip_fail(e, "fork"));
	rc = blake2_dev(c, nup, st.d_big, h_soff, h_slen, 0, 0, st.h_buf + ssum_off, st.stream3, 0, 0, 0, true, S);
	if (rc)
		return finish(rc);
	e = hipEventRecord(ev_sh, st.stream3);
	if (e != hipSuccess)
		return finish(hip_fail(e, "hipEventRecord"));
	if (!staged) {
		// one stage: decode per bucket, then the block checksums, then the rebuilt shards go home, all on the main stream
		std::vector<gec::CopyEntry> outs;
		for (auto &kv : buckets)
			if (kv.second.npar) {
				rc = decode_bucket(kv.second, st.stream, outs);
				if (rc)
					return finish(rc);
			}
		if (block_sums) {
			rc = blake2_dev(c, nblocks, st.d_big, h_boff, h_blen, 0, 0, st.h_buf + bsum_off, st.stream);
			if (rc)
				return finish(rc);
		}
		rc = launch_copy_table(st, outs, st.stream);
		if (rc)
			return finish(rc);
	} else {
		e = hipEventRecord(st.ev_join, chain_stream);
		if (e == hipSuccess)
			e = hipStreamWaitEvent(st.stream, st.ev_join, 0);
		if (e == hipSuccess && down_stream != st.stream) {
			e = hipEventRecord(st.ev_out, down_stream);
			if (e == hipSuccess)
				e = hipStreamWaitEvent(st.stream, st.ev_out, 0);
		}
		if (e != hipSuccess)
			return finish(hip_fail(e, "join"));
	}
	if (link_release_armed() && nreb == 0 && hipEventSynchronize(ev_up) == hipSuccess)
		link_release_fire();  // everything is up and nothing goes home: the link is free while the checksums run
	e = hipStreamWaitEvent(st.stream, ev_sh, 0);
	if (e == hipSuccess)
		e = hipStreamSynchronize(st.stream);
	if (e != hipSuccess)
		return finish(hip_fail(e, "hipStreamSynchronize"));
	// -- results: checksums to where the caller indexes them, rebuilt shards that could not be written directly
	for (size_t i = 0; i < ups.size(); ++i)
		std::memcpy(shard_sums + 32 * ups[i].idx, st.h_buf + ssum_off + 32 * i, 32);
	if (block_sums)
		for (size_t i = 0; i < nblocks; ++i)
			std::memcpy(block_sums + 32 * block_order[i], st.h_buf + bsum_off + 32 * i, 32);
	for (auto &kv : buckets) {
		Bucket &bk = kv.second;
		size_t w = 0;
		for (size_t i = 0; i < bk.ids.size(); ++i)
			for (size_t r = 0; r < bk.npar; ++r, ++w) {
				uint8_t *dst = rebuilt[bk.ids[i] * n + bk.plan->missing[r]];
				if (!(aligned16(dst) && pinned().contains(dst, S)))
					std::memcpy(dst, st.h_buf + reb_off + bk.rq[w] * S, S);
			}
	}
	return finish(GEC_OK);
}

int HipBackend::verify_batch(size_t nblocks, const uint8_t *const *shards, size_t S, uint8_t *ok)
{
	const size_t n = c->k + c->m;
	const size_t stripe = n * S;
	ForegroundScope fg(c);
	bool all_pinned = (size_t)c->k <= (size_t)gec::PTR_KMAX;
	for (size_t i = 0; i < nblocks * n && all_pinned; ++i)
		all_pinned = aligned16(shards[i]) && pinned().contains(shards[i], S);
	if (all_pinned) {
		// scrub of shards that sit in pinned memory: one kernel reads all k+m shards over the link and leaves the
		// per-block verdicts in pinned memory; nothing is staged
		const size_t k = c->k, m = c->m;
		DeviceGuard dg(c->device);
		if (!dg.ok)
			return fail(GEC_E_DEVICE, "hipSetDevice failed");
		StagingLease lease(c);
		Staging &st = lease.st;
		int rc = st.ensure(64, nblocks);
		if (!rc)
			rc = st.ensure_tab((nblocks * (k * 12 + m * 8) + 64) / sizeof(gec::CopyEntry) + 8);
		if (rc)
			return rc;
		std::vector<const uint8_t *> in(nblocks * k);
		std::vector<uint32_t> valid(nblocks * k, (uint32_t)S);
		std::vector<uint8_t *> par(nblocks * m);
		for (size_t b = 0; b < nblocks; ++b) {
			for (size_t t = 0; t < k; ++t)
				in[b * k + t] = pinned().dev(shards[b * n + t]);
			for (size_t r = 0; r < m; ++r)
				par[b * m + r] = const_cast<uint8_t *>(pinned().dev(shards[b * n + k + r]));
		}
		std::memset(st.h_bad, 0, nblocks * sizeof(uint32_t));
		rc = launch_apply_ptrs(c, st, nblocks, in.data(), valid.data(), par.data(), (int)m, S, c->enc.row((int)k), st.stream, nullptr, st.h_bad);
		const hipError_t e = hipStreamSynchronize(st.stream);
		if (rc)
			return rc;
		HIP_TRY(e);
		for (size_t b = 0; b < nblocks; ++b)
			ok[b] = st.h_bad[b] ? 0 : 1;
		return GEC_OK;
	}
	const size_t ch = chunk_blocks(stripe, nblocks, kChunkBytes);
	ForkJoinPool &pool = copy_pool();
	return run_pipeline(
		c, (nblocks + ch - 1) / ch, ch * stripe, ch,
		[&](size_t ci, Staging &st) {
			const size_t b0 = ci * ch, nb = std::min(ch, nblocks - b0);
			pool.parallel_for(nb * n, [&](size_t q) {
				std::memcpy(st.h_buf + (q / n) * stripe + (q % n) * S, shards[(b0 + q / n) * n + q % n], S);
			});
		},
		[&](size_t ci, Staging &st) -> int {
			const size_t nb = std::min(ch, nblocks - ci * ch);
			HIP_TRY(hipMemcpyAsync(st.d_buf, st.h_buf, nb * stripe, hipMemcpyHostToDevice, st.stream));
			int rc = verify_dev(c, nb, st.d_buf, stripe, S, st.d_bad, st.stream);
			if (rc)
				return rc;
			HIP_TRY(hipMemcpyAsync(st.h_bad, st.d_bad, nb * sizeof(uint32_t), hipMemcpyDeviceToHost, st.stream));
			return GEC_OK;
		},
		[&](size_t ci, Staging &st) {
			const size_t b0 = ci * ch, nb = std::min(ch, nblocks - b0);
			for (size_t i = 0; i < nb; ++i)
				ok[b0 + i] = st.h_bad[i] ? 0 : 1;
		});
}

int HipBackend::verify_hash_batch(size_t nblocks, const uint8_t *const *shards, size_t S, uint8_t *ok, uint8_t *shard_sums)
{
	const size_t k = c->k, m = c->m, n = k + m;
	ForegroundScope fg(c);
	bool all_pinned = k <= (size_t)gec::PTR_KMAX;
	for (size_t i = 0; i < nblocks * n && all_pinned; ++i)
		all_pinned = aligned16(shards[i]) && pinned().contains(shards[i], S);
	if (!all_pinned) {
		// pageable shards: two staged trips (the scrub of shards a caller keeps in ordinary memory is not a hot path)
		int rc = verify_batch(nblocks, shards, S, ok);
		if (rc)
			return rc;
		std::vector<size_t> lens(nblocks * n, S);
		return hash_batch(nblocks * n, shards, lens.data(), shard_sums, true);
	}
	// one trip: the compare form of gf_apply_ptrs reads all k+m shards of a chunk over the link, leaves the verdicts in
	// pinned memory and everything it read in HBM, where the chunk's shard checksums are computed while the next
	// chunk is on the link (same stream pair as gec_encode_hash_batch)
	DeviceGuard dg(c->device);
	if (!dg.ok)
		return fail(GEC_E_DEVICE, "hipSetDevice failed");
	StagingLease lease(c);
	Staging &st = lease.st;
	const size_t stripe = n * S;
	const size_t zch = chunk_blocks(stripe, nblocks, trip_chunk_bytes(c));
	const size_t nz = (nblocks + zch - 1) / zch;
	int rc = st.ensure(nblocks * n * 32 + 64, nblocks);
	if (!rc)
		rc = st.ensure_tab((nblocks * (k * 12 + m * 8) + 64 * nz) / sizeof(gec::CopyEntry) + 4 * nz + 4);
	if (!rc && c->sumkind != GEC_SHARDSUM_MLH64)
		rc = st.ensure_big(2 * (zch * stripe + zch * n * 32));
	if (!rc)
		rc = st.ensure_segments(num_cu);
	if (rc)
		return rc;
	hipStream_t up = st.stream_up ? st.stream_up : st.stream;
	hipStream_t chain = st.stream_chain ? st.stream_chain : st.stream2;
	std::memset(st.h_bad, 0, nblocks * sizeof(uint32_t));
	std::vector<const uint8_t *> in(zch * k);
	std::vector<uint32_t> valid(zch * k, (uint32_t)S);
	std::vector<uint8_t *> par(zch * m);
	// checksum kind 3: the compare form of the link kernel also leaves the leaf sums of the k shards it reads and of the m STORED rows
	// it checks, from its registers: nothing is mirrored in HBM, the roots follow on the same stream
	const bool v3 = c->sumkind == GEC_SHARDSUM_MLH64;
	const uint32_t nleaf_max = (uint32_t)((S + gec::SHARDSUM_LEAF - 1) / gec::SHARDSUM_LEAF);
	uint8_t *scr = nullptr;
	if (v3)
		rc = gecimpl::leaf_scratch(c, up, zch * n * nleaf_max * 8, &scr);
	for (size_t ci = 0; ci < nz && !rc; ++ci) {
		const size_t b0 = ci * zch, nb = std::min(zch, nblocks - b0);
		background_yield(c);
		for (size_t i = 0; i < nb; ++i) {
			for (size_t t = 0; t < k; ++t)
				in[i * k + t] = pinned().dev(shards[(b0 + i) * n + t]);
			for (size_t r = 0; r < m; ++r)
				par[i * m + r] = const_cast<uint8_t *>(pinned().dev(shards[(b0 + i) * n + k + r]));
		}
		if (v3) {
			const SumOut so{reinterpret_cast<uint64_t *>(scr), nleaf_max, (uint32_t)n, 0u, true};
			rc = launch_apply_ptrs(c, st, nb, in.data(), valid.data(), par.data(), (int)m, S, c->enc.row((int)k), up, nullptr, st.h_bad + b0, 0, nullptr, &so);
			if (!rc)
				rc = mlh_roots_dev(c, nb * n, so.lsum, nleaf_max, nullptr, S, st.h_buf + b0 * n * 32, up);
			continue;
		}
		uint8_t *mir = st.d_big + (ci & 1) * (zch * stripe + zch * n * 32);
		if (ci >= 2 && hipStreamWaitEvent(up, st.ev_seg[2 + (ci & 1)], 0) != hipSuccess)
			rc = fail(GEC_E_DEVICE, "hipStreamWaitEvent");
		if (!rc)
			rc = launch_apply_ptrs(c, st, nb, in.data(), valid.data(), par.data(), (int)m, S, c->enc.row((int)k), up, mir, st.h_bad + b0);
		if (rc)
			continue;
		hipError_t e = hipEventRecord(st.ev_seg[ci & 1], up);
		if (e == hipSuccess)
			e = hipStreamWaitEvent(chain, st.ev_seg[ci & 1], 0);
		if (e != hipSuccess) {
			rc = fail(GEC_E_DEVICE, "chunk event");
			continue;
		}
		rc = blake2_dev(c, nb * n, mir, nullptr, nullptr, S, S, st.h_buf + b0 * n * 32, chain, 0, 0, 0, true);
		if (!rc && hipEventRecord(st.ev_seg[2 + (ci & 1)], chain) != hipSuccess)
			rc = fail(GEC_E_DEVICE, "hipEventRecord");
	}
	const hipError_t e1 = hipStreamSynchronize(up), e2 = hipStreamSynchronize(chain);
	if (rc)
		return rc;
	HIP_TRY(e1);
	HIP_TRY(e2);
	std::memcpy(shard_sums, st.h_buf, nblocks * n * 32);
	for (size_t b = 0; b < nblocks; ++b)
		ok[b] = st.h_bad[b] ? 0 : 1;
	return GEC_OK;
}

int HipBackend::reconstruct_batch(size_t nblocks, const uint8_t *const *shards, uint8_t *const *out, size_t S, int data_only,
				  uint8_t *in_sums, uint8_t *out_sums)
{
	const size_t k = c->k, n = c->k + c->m;
	ForegroundScope fg(c);
	// bucket blocks by erasure pattern AND by which of the missing shards the caller wants back
	// (out entry non-NULL; with data_only parity is never wanted): one decode plan per bucket, one
	// launch per chunk, only the wanted rows computed.  key[j]: 1 present, 0 missing+wanted, 2 missing+unwanted
	std::map<std::string, std::vector<size_t>> buckets;
	for (size_t b = 0; b < nblocks; ++b) {
		std::string key(n, 0);
		size_t npresent = 0, nwanted = 0;
		for (size_t j = 0; j < n; ++j) {
			if (shards[b * n + j]) {
				key[j] = 1;
				++npresent;
			} else if ((data_only && j >= k) || !out[b * n + j]) {
				key[j] = 2;
			} else {
				++nwanted;
			}
		}
		if (npresent < k)
			return fail(GEC_E_TOO_FEW_PRESENT, "fewer than k shards present");
		if (nwanted)
			buckets[key].push_back(b);
	}
	ForkJoinPool &pool = copy_pool();
	// one decode plan per bucket, cut down to the rows the caller wants
	struct Work {
		const std::vector<size_t> *ids;
		std::shared_ptr<const Plan> plan;
		bool all_pinned;
	};
	std::vector<Work> work;
	bool every_pinned = true;
	size_t tab_bytes = 0;
	for (auto &kv : buckets) {
		const std::vector<size_t> &ids = kv.second;
		std::string pres(kv.first);
		for (auto &ch : pres)
			ch = ch == 1 ? 1 : 0;
		std::shared_ptr<const Plan> full;
		int rc = get_plan(c, reinterpret_cast<const uint8_t *>(pres.data()), false, full);
		if (rc)
			return rc;
		auto sub = std::make_shared<Plan>();
		sub->valid = full->valid;
		for (size_t r = 0; r < full->missing.size(); ++r)
			if (kv.first[full->missing[r]] == 0)
				sub->missing.push_back(full->missing[r]);
		sub->rows = gec::Matrix((int)sub->missing.size(), (int)k);
		for (size_t r = 0, w = 0; r < full->missing.size(); ++r)
			if (kv.first[full->missing[r]] == 0)
				std::memcpy(&sub->rows.at((int)w++, 0), full->rows.row((int)r), k);
		const size_t nmiss = sub->missing.size();
		if (nmiss == 0)
			continue;
		bool all_pinned = true;
		for (size_t i = 0; i < ids.size() && all_pinned; ++i) {
			for (size_t t = 0; t < k && all_pinned; ++t)
				all_pinned = aligned16(shards[ids[i] * n + sub->valid[t]]) && pinned().contains(shards[ids[i] * n + sub->valid[t]], S);
			for (size_t r = 0; r < nmiss && all_pinned; ++r)
				all_pinned = aligned16(out[ids[i] * n + sub->missing[r]]) && pinned().contains(out[ids[i] * n + sub->missing[r]], S);
		}
		every_pinned = every_pinned && all_pinned;
		tab_bytes += ids.size() * (k * 12 + nmiss * 8) + 64;
		work.push_back({&ids, sub, all_pinned});
	}
	if (!work.empty() && every_pinned && k <= (size_t)gec::PTR_KMAX) {
		// every shard and every output is device-addressable: one gf_apply_ptrs launch per erasure pattern reads
		// the k shards the decode uses and writes the rebuilt ones straight over the link
		DeviceGuard dg(c->device);
		if (!dg.ok)
			return fail(GEC_E_DEVICE, "hipSetDevice failed");
		StagingLease lease(c);
		Staging &st = lease.st;
		const bool sums = in_sums != nullptr;
		// with checksums: chunks of every pattern run through the two mirror halves in turn (chunk q+2 waits for
		// the checksums of chunk q), link kernels on the upload CUs, checksum kernels on the rest
		size_t sum_bytes = 0, max_ids = 0, nchunks_total = 0;
		const size_t zch_cap = chunk_blocks(n * S, nblocks, trip_chunk_bytes(c));
		for (const Work &w : work) {
			sum_bytes += w.ids->size() * (k + w.plan->missing.size()) * 32;
			max_ids = std::max(max_ids, w.ids->size());
			nchunks_total += (w.ids->size() + zch_cap - 1) / zch_cap;
		}
		const size_t zch = std::min(zch_cap, std::max<size_t>(max_ids, 1));
		const size_t half = zch * n * S + zch * n * 32;
		// checksum kind 3: the link kernel leaves the leaf sums of what it reads and writes itself (SUM form): no mirror, one stream
		const bool v3 = sums && c->sumkind == GEC_SHARDSUM_MLH64;
		const uint32_t nleaf_max = (uint32_t)((S + gec::SHARDSUM_LEAF - 1) / gec::SHARDSUM_LEAF);
		int rc = st.ensure(sums ? sum_bytes + 64 : 64, 0);
		if (!rc)
			rc = st.ensure_tab(tab_bytes / sizeof(gec::CopyEntry) + 4 * (work.size() + nchunks_total) + 4);
		if (!rc && sums && !v3)
			rc = st.ensure_big(2 * half);
		if (!rc && sums)
			rc = st.ensure_segments(num_cu);
		if (rc)
			return rc;
		hipStream_t up = sums && st.stream_up ? st.stream_up : st.stream;
		hipStream_t chain = sums && st.stream_chain ? st.stream_chain : st.stream2;
		uint8_t *scr = nullptr;
		if (v3)
			rc = gecimpl::leaf_scratch(c, up, zch * n * nleaf_max * 8, &scr);
		size_t q = 0, sum_off = 0;  // running chunk number, running offset into the pinned checksum area
		std::vector<size_t> sum_base(work.size());
		for (size_t wi = 0; wi < work.size() && !rc; ++wi) {
			const Work &w = work[wi];
			const std::vector<size_t> &ids = *w.ids;
			const size_t nmiss = w.plan->missing.size(), per = k + nmiss;
			sum_base[wi] = sum_off;
			std::vector<const uint8_t *> in(std::min(zch, ids.size()) * k);
			std::vector<uint32_t> valid(in.size(), (uint32_t)S);
			std::vector<uint8_t *> outp(std::min(zch, ids.size()) * nmiss);
			const bool chunked = sums || c->qos_class == GEC_CLASS_BACKGROUND;  // a background codec always goes in chunks
			for (size_t i0 = 0; i0 < ids.size() && !rc; i0 += chunked ? zch : ids.size(), ++q) {
				const size_t nb = chunked ? std::min(zch, ids.size() - i0) : ids.size();
				background_yield(c);
				if (!sums) {
					in.resize(nb * k);
					valid.assign(nb * k, (uint32_t)S);
					outp.resize(nb * nmiss);
				}
				for (size_t i = 0; i < nb; ++i) {
					for (size_t t = 0; t < k; ++t)
						in[i * k + t] = pinned().dev(shards[ids[i0 + i] * n + w.plan->valid[t]]);
					for (size_t r = 0; r < nmiss; ++r)
						outp[i * nmiss + r] = pinned().dev(out[ids[i0 + i] * n + w.plan->missing[r]]);
				}
				if (v3) {
					const SumOut so{reinterpret_cast<uint64_t *>(scr), nleaf_max, (uint32_t)per, 0u, true};
					rc = launch_apply_ptrs(c, st, nb, in.data(), valid.data(), outp.data(), (int)nmiss, S, w.plan->rows.v.data(), up, nullptr, nullptr, 0,
							       nullptr, &so);
					if (!rc)
						rc = mlh_roots_dev(c, nb * per, so.lsum, nleaf_max, nullptr, S, st.h_buf + sum_off, up);
					sum_off += nb * per * 32;
					continue;
				}
				uint8_t *mir = sums ? st.d_big + (q & 1) * half : nullptr;
				if (sums && q >= 2 && hipStreamWaitEvent(up, st.ev_seg[2 + (q & 1)], 0) != hipSuccess)
					rc = fail(GEC_E_DEVICE, "hipStreamWaitEvent");
				if (!rc)
					rc = launch_apply_ptrs(c, st, nb, in.data(), valid.data(), outp.data(), (int)nmiss, S, w.plan->rows.v.data(), up, mir);
				if (rc || !sums)
					continue;
				hipError_t e = hipEventRecord(st.ev_seg[q & 1], up);
				if (e == hipSuccess)
					e = hipStreamWaitEvent(chain, st.ev_seg[q & 1], 0);
				if (e != hipSuccess) {
					rc = fail(GEC_E_DEVICE, "chunk event");
					continue;
				}
				rc = blake2_dev(c, nb * per, mir, nullptr, nullptr, S, S, st.h_buf + sum_off, chain, 0, 0, 0, true);
				if (!rc && hipEventRecord(st.ev_seg[2 + (q & 1)], chain) != hipSuccess)
					rc = fail(GEC_E_DEVICE, "hipEventRecord");
				sum_off += nb * per * 32;
			}
		}
		const hipError_t e1 = hipStreamSynchronize(up), e2 = sums ? hipStreamSynchronize(chain) : hipSuccess;  // also on error: queued launches read the tables
		if (rc)
			return rc;
		HIP_TRY(e1);
		HIP_TRY(e2);
		if (sums)
			for (size_t wi = 0; wi < work.size(); ++wi) {
				const Work &w = work[wi];
				const size_t nmiss = w.plan->missing.size(), per = k + nmiss;
				for (size_t i = 0; i < w.ids->size(); ++i) {
					const uint8_t *src = st.h_buf + sum_base[wi] + i * per * 32;
					const size_t b = (*w.ids)[i];
					for (size_t t = 0; t < k; ++t)
						std::memcpy(in_sums + 32 * (b * n + w.plan->valid[t]), src + 32 * t, 32);
					for (size_t r = 0; r < nmiss; ++r)
						std::memcpy(out_sums + 32 * (b * n + w.plan->missing[r]), src + 32 * (k + r), 32);
				}
			}
		return GEC_OK;
	}
	if (in_sums) {
		// buffers the device cannot address: the staged reconstruct, then the checksums of what was read and written in
		// a second trip
		int rc = reconstruct_batch(nblocks, shards, out, S, data_only, nullptr, nullptr);
		if (rc)
			return rc;
		std::vector<const uint8_t *> msgs;
		std::vector<size_t> lens, where;
		std::vector<uint8_t *> dst;
		for (const Work &w : work)
			for (size_t b : *w.ids) {
				for (size_t t = 0; t < k; ++t) {
					msgs.push_back(shards[b * n + w.plan->valid[t]]);
					dst.push_back(in_sums + 32 * (b * n + w.plan->valid[t]));
				}
				for (int j : w.plan->missing) {
					msgs.push_back(out[b * n + j]);
					dst.push_back(out_sums + 32 * (b * n + j));
				}
			}
		lens.assign(msgs.size(), S);
		std::vector<uint8_t> tmp(msgs.size() * 32);
		rc = hash_batch(msgs.size(), msgs.data(), lens.data(), tmp.data(), true);
		for (size_t i = 0; !rc && i < msgs.size(); ++i)
			std::memcpy(dst[i], tmp.data() + 32 * i, 32);
		return rc;
	}
	for (const Work &wk : work) {
		const std::vector<size_t> &ids = *wk.ids;
		std::shared_ptr<const Plan> plan = wk.plan;
		const size_t nmiss = plan->missing.size();
		const size_t stripe = (k + nmiss) * S;
		const bool all_pinned = wk.all_pinned;
		int rc = GEC_OK;
		const size_t ch = chunk_blocks(stripe, ids.size(), all_pinned ? pinned_chunk_bytes(c) : kChunkBytes);
		std::vector<size_t> in_off(k), out_off(nmiss);
		for (size_t t = 0; t < k; ++t)
			in_off[t] = t * S;
		for (size_t r = 0; r < nmiss; ++r)
			out_off[r] = (k + r) * S;
		// chunks whose input shards (resp. output buffers) all lie in registered pinned memory go by
		// DMA straight from / to the caller's memory; adjacent shards (slices of one block buffer)
		// are merged into one copy
		const size_t nchunks = (ids.size() + ch - 1) / ch;
		std::vector<uint8_t> in_pinned(nchunks, 1), out_pinned(nchunks, 1);
		PipeChain chain;
		for (size_t i = 0; i < ids.size(); ++i) {
			const size_t ci = i / ch;
			for (size_t t = 0; t < k && in_pinned[ci]; ++t)
				if (!(aligned16(shards[ids[i] * n + plan->valid[t]]) && pinned().contains(shards[ids[i] * n + plan->valid[t]], S)))
					in_pinned[ci] = 0;
			for (size_t r = 0; r < nmiss && out_pinned[ci]; ++r)
				if (!(aligned16(out[ids[i] * n + plan->missing[r]]) && pinned().contains(out[ids[i] * n + plan->missing[r]], S)))
					out_pinned[ci] = 0;
		}
		rc = run_pipeline(
			c, nchunks, ch * stripe, 0,
			[&](size_t ci, Staging &st) {
				(void)st.ensure_tab(ch * (k + nmiss));
				if (in_pinned[ci])
					return;
				const size_t i0 = ci * ch, nb = std::min(ch, ids.size() - i0);
				pool.parallel_for(nb * k, [&](size_t q) {
					const size_t i = q / k, t = q % k;
					std::memcpy(st.h_buf + i * stripe + t * S, shards[ids[i0 + i] * n + plan->valid[t]], S);
				});
			},
			[&](size_t ci, Staging &st) -> int {
				const size_t i0 = ci * ch, nb = std::min(ch, ids.size() - i0);
				if (in_pinned[ci]) {
					std::vector<gec::CopyEntry> ents;
					ents.reserve(nb * 3);
					for (size_t i = 0; i < nb; ++i) {
						const uint8_t *const *sh = shards + ids[i0 + i] * n;
						for (size_t t = 0; t < k;) {  // adjacent shards (slices of one block buffer): one entry
							size_t run = 1;
							while (t + run < k && sh[plan->valid[t + run]] == sh[plan->valid[t]] + run * S)
								++run;
							ents.push_back({pinned().dev(sh[plan->valid[t]]), st.d_buf + i * stripe + t * S, run * S});
							t += run;
						}
					}
					int rct = chain.before(chain.last_in, st.stream);
					if (!rct)
						rct = launch_copy_table(st, ents, st.stream);
					if (!rct)
						rct = chain.after_in(st);
					if (rct)
						return rct;
				} else {
					HIP_TRY(hipMemcpy2DAsync(st.d_buf, stripe, st.h_buf, stripe, k * S, nb, hipMemcpyHostToDevice, st.stream));
				}
				int r2 = launch_apply(c, st.d_buf, stripe, st.d_buf, stripe, nullptr, 0, S, nb, in_off.data(),
						      out_off.data(), (int)nmiss, plan->rows.v.data(), gec::MODE_STORE, st.stream);
				if (r2)
					return r2;
				if (out_pinned[ci]) {
					std::vector<gec::CopyEntry> ents;
					ents.reserve(nb * nmiss);
					for (size_t i = 0; i < nb; ++i) {
						uint8_t *const *o = out + ids[i0 + i] * n;
						for (size_t r = 0; r < nmiss; ++r)
							ents.push_back({st.d_buf + i * stripe + (k + r) * S, pinned().dev(o[plan->missing[r]]), S});
					}
					int rct = chain.before(chain.last_out, st.stream);
					if (!rct)
						rct = launch_copy_table(st, ents, st.stream);
					if (!rct)
						rct = chain.after_out(st);
					if (rct)
						return rct;
				} else {
					HIP_TRY(hipMemcpy2DAsync(st.h_buf + k * S, stripe, st.d_buf + k * S, stripe, nmiss * S, nb, hipMemcpyDeviceToHost, st.stream));
				}
				return GEC_OK;
			},
			[&](size_t ci, Staging &st) {
				if (out_pinned[ci])
					return;
				const size_t i0 = ci * ch, nb = std::min(ch, ids.size() - i0);
				pool.parallel_for(nb * nmiss, [&](size_t q) {
					const size_t i = q / nmiss, r = q % nmiss;
					std::memcpy(out[ids[i0 + i] * n + plan->missing[r]], st.h_buf + i * stripe + (k + r) * S, S);
				});
			});
		if (rc)
			return rc;
	}
	return GEC_OK;
}

}  // namespace gecimpl
