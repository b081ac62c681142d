//! Emits reference vectors for the batched authenticated-share path of ark-mpc as JSON on stdout.
//! Everything is computed by ark-mpc / arkworks 0.4 themselves; this file only chooses inputs and prints.
//! See README.md for how the vectors are consumed (tests/test_ref_vectors.py).

use std::future::Future;
use std::marker::PhantomData;

use ark_bls12_381::G1Projective as Bls12_381;
use ark_bn254::G1Projective as Bn254;
use ark_curve25519::EdwardsProjective as Ed25519;
use ark_ec::CurveGroup;
use ark_mpc::{
    algebra::{AuthenticatedPointResult, AuthenticatedScalarResult, CurvePoint, PointShare, Scalar, ScalarShare},
    network::{MockNetwork, NetworkOutbound, NetworkPayload, UnboundedDuplexStream},
    offline_prep::PreprocessingPhase,
    test_helpers::execute_mock_mpc,
    MpcFabric, PARTY0, PARTY1,
};
use num_bigint::BigUint;
use serde_json::{json, Value};
use sha3::{Digest, Sha3_256};

fn hex(bytes: &[u8]) -> String {
    bytes.iter().map(|b| format!("{b:02x}")).collect()
}

fn dec<C: CurveGroup>(s: &Scalar<C>) -> String {
    s.to_biguint().to_string()
}

fn scalar_from_dec<C: CurveGroup>(d: &str) -> Scalar<C> {
    Scalar::from_biguint(&d.parse::<BigUint>().unwrap())
}

/// Edge and generic values: 0, 1, 2, r-1, r-2, 2^64, 2^128 + 12345, 2^192 - 1, 2^250 + 7 and a few fixed words (all reduced mod r)
fn test_scalars<C: CurveGroup>() -> Vec<Scalar<C>> {
    let minus_one = Scalar::<C>::from(0u8) - Scalar::<C>::from(1u8);
    let mut v = vec![
        Scalar::from(0u8),
        Scalar::from(1u8),
        Scalar::from(2u8),
        minus_one,
        minus_one - Scalar::from(1u8),
        Scalar::from_biguint(&(BigUint::from(1u8) << 64)),
        Scalar::from_biguint(&((BigUint::from(1u8) << 128) + BigUint::from(12345u32))),
        Scalar::from_biguint(&((BigUint::from(1u8) << 192) - BigUint::from(1u8))),
        Scalar::from_biguint(&((BigUint::from(1u8) << 250) + BigUint::from(7u8))),
    ];
    // a deterministic tail: x_{k+1} = x_k^2 + 0x9E3779B97F4A7C15
    let mut x = Scalar::<C>::from(0xA11CE001u64);
    for _ in 0..8 {
        x = x * x + Scalar::from(0x9E3779B97F4A7C15u64);
        v.push(x);
    }
    v
}

fn scalar_vectors<C: CurveGroup>(curve: &str) -> Vec<Value> {
    test_scalars::<C>()
        .iter()
        .map(|s| {
            json!({
                "curve": curve,
                "value_dec": dec(s),
                "to_bytes_be_hex": hex(&s.to_bytes_be()),                      // scalar.rs:118-127
                "serde_json": String::from_utf8(serde_json::to_vec(s).unwrap()).unwrap(), // scalar.rs:186-192: 32 LE bytes as numbers
            })
        })
        .collect()
}

fn point_vectors<C: CurveGroup>(curve: &str) -> Vec<Value> {
    let g = CurvePoint::<C>::generator();
    test_scalars::<C>()
        .iter()
        .map(|k| {
            let p = g * *k;                                                   // curve.rs:403-409
            json!({
                "curve": curve,
                "scalar_dec": dec(k),
                "to_bytes_hex": hex(&p.to_bytes()),                           // curve.rs:103-108 (serialize_compressed)
                "neg_to_bytes_hex": hex(&(-p).to_bytes()),
                "double_to_bytes_hex": hex(&(p + p).to_bytes()),
            })
        })
        .collect()
}

fn wire_vectors<C: CurveGroup>(curve: &str, with_points: bool) -> Vec<Value> {
    let scalars = test_scalars::<C>();
    let g = CurvePoint::<C>::generator();
    let points: Vec<CurvePoint<C>> = scalars.iter().take(5).map(|k| g * *k).collect();
    let mut out = Vec::new();
    for (result_id, n) in [(6usize, 0usize), (7, 1), (1234567, scalars.len())] {
        let msg = NetworkOutbound::<C> { result_id, payload: NetworkPayload::ScalarBatch(scalars[..n].to_vec()) };
        out.push(json!({
            "curve": curve, "variant": "ScalarBatch", "result_id": result_id,
            "values_dec": scalars[..n].iter().map(dec).collect::<Vec<_>>(),
            "json": String::from_utf8(serde_json::to_vec(&msg).unwrap()).unwrap(),   // network.rs:33-60; quic.rs:303-306 prefixes a u64 LE length
        }));
    }
    if !with_points {
        return out; // (BLS12-381 is on the engine's path as a scalar field only: config 5)
    }
    let msg = NetworkOutbound::<C> { result_id: 99, payload: NetworkPayload::PointBatch(points.clone()) };
    out.push(json!({
        "curve": curve, "variant": "PointBatch", "result_id": 99,
        "scalars_dec": scalars.iter().take(5).map(dec).collect::<Vec<_>>(),
        "points_to_bytes_hex": points.iter().map(|p| hex(&p.to_bytes())).collect::<Vec<_>>(),
        "json": String::from_utf8(serde_json::to_vec(&msg).unwrap()).unwrap(),
    }));
    out
}

/// commitment.rs:71-86 with the public methods it calls: SHA3-256 over to_bytes_be(v_0) || ... || to_bytes_be(blinder),
/// reduced with from_be_bytes_mod_order (HashCommitment itself is pub(crate))
fn commitment_vectors<C: CurveGroup>(curve: &str) -> Vec<Value> {
    let scalars = test_scalars::<C>();
    let mut out = Vec::new();
    for n in [0usize, 1, 4, scalars.len()] {
        let blinder = scalars[scalars.len() - 1 - (n % 3)];
        let mut hasher = Sha3_256::new();
        for v in &scalars[..n] {
            hasher.update(v.to_bytes_be());
        }
        hasher.update(blinder.to_bytes_be());
        let digest = hasher.finalize();
        let commitment = Scalar::<C>::from_be_bytes_mod_order(&digest);
        out.push(json!({
            "curve": curve,
            "values_dec": scalars[..n].iter().map(dec).collect::<Vec<_>>(),
            "blinder_dec": dec(&blinder),
            "sha3_256_hex": hex(&digest),
            "commitment_dec": dec(&commitment),
        }));
    }
    out
}

/// AuthenticatedScalarResult::batch_mul (authenticated_scalar.rs:848-879) under execute_mock_mpc with PartyIDBeaverSource: each party's
/// local ScalarShare of the shared inputs and of every product, and the authenticated opening (bn254: the reference's TestCurve)
async fn batch_mul_vectors() -> Value {
    let xs: Vec<String> = test_scalars::<Bn254>().iter().map(dec).collect();
    let ys: Vec<String> = test_scalars::<Bn254>().iter().rev().map(dec).collect();
    let (xs2, ys2) = (xs.clone(), ys.clone());
    let (p0, p1) = execute_mock_mpc(move |fabric| {
        let (xs, ys) = (xs2.clone(), ys2.clone());
        async move {
            let x: Vec<Scalar<Bn254>> = xs.iter().map(|d| scalar_from_dec(d)).collect();
            let y: Vec<Scalar<Bn254>> = ys.iter().map(|d| scalar_from_dec(d)).collect();
            let a = fabric.batch_share_scalar(x, PARTY0);                      // fabric.rs:578-600
            let b = fabric.batch_share_scalar(y, PARTY0);
            let prod = AuthenticatedScalarResult::batch_mul(&a, &b);
            let mut rows = Vec::new();
            for (i, p) in prod.iter().enumerate() {
                let (sa, sb, sp) = (a[i].clone().await, b[i].clone().await, p.clone().await);   // this party's ScalarShare values
                rows.push(json!({
                    "x_share": [dec(&sa.share()), dec(&sa.mac())],
                    "y_share": [dec(&sb.share()), dec(&sb.mac())],
                    "product_share": [dec(&sp.share()), dec(&sp.mac())],
                }));
            }
            let opened = AuthenticatedScalarResult::open_authenticated_batch(&prod);
            let mut vals = Vec::new();
            for o in opened {
                vals.push(dec(&o.await.expect("MAC check")));
            }
            json!({ "shares": rows, "opened_dec": vals })
        }
    })
    .await;
    json!({ "curve": "bn254", "source": "PartyIDBeaverSource", "x_dec": xs, "y_dec": ys, "party0": p0, "party1": p1 })
}

// ---------------------------------------------------------------------------------------------------------------------------------
// A NON-degenerate preprocessing source, generic over the curve.  PartyIDBeaverSource gives party 0 the MAC key share 0 and every value a
// constant (offline_prep.rs:103-170), so under it party 0's MACs are all zero and every gate sees the same triple.  FixedSource hands out
// index-dependent values with index-dependent additive splits under two non-trivial key shares; everything is a closed formula of the
// running index, restated in tools/ref_vectors/model_vectors.py:
//     key share of party 0 / 1 = 0x1234567 / 0x89abcdef,  key = their sum
//     val(tag, i) = ((tag << 32) + i + 1)^3 + 0x9E3779B97F4A7C15                      (in the scalar field)
//     a value v with index i under tag t is split as  party 0: (r, m)   party 1: (v - r, key * v - m),  r = val(t + 16, i), m = val(t + 32, i)
//     triple i: a = val(1, i), b = val(2, i), c = a * b;  input mask i = val(3, i) (the sender knows it in the clear);
//     shared bit i = i & 1;  shared value i = val(5, i);  inverse pair i = (val(6, i), val(6, i)^-1)
// ---------------------------------------------------------------------------------------------------------------------------------
struct FixedSource<C: CurveGroup> {
    party: u64,
    ctr: [u64; 7],
    _c: PhantomData<fn() -> C>,
}
impl<C: CurveGroup> FixedSource<C> {
    fn new(party: u64) -> Self {
        Self { party, ctr: [0; 7], _c: PhantomData }
    }
    fn key_share(party: u64) -> Scalar<C> {
        Scalar::from(if party == 0 { 0x1234567u64 } else { 0x89abcdefu64 })
    }
    fn key() -> Scalar<C> {
        Self::key_share(0) + Self::key_share(1)
    }
    fn val(tag: u64, i: u64) -> Scalar<C> {
        let b = Scalar::<C>::from((tag << 32) + i + 1);
        b * b * b + Scalar::from(0x9E3779B97F4A7C15u64)
    }
    fn split(&self, tag: u64, i: u64, v: Scalar<C>) -> ScalarShare<C> {
        let (r, m) = (Self::val(tag + 16, i), Self::val(tag + 32, i));
        if self.party == 0 {
            ScalarShare::new(r, m)
        } else {
            ScalarShare::new(v - r, Self::key() * v - m)
        }
    }
    fn next(&mut self, tag: usize) -> u64 {
        let i = self.ctr[tag];
        self.ctr[tag] += 1;
        i
    }
}
impl<C: CurveGroup> PreprocessingPhase<C> for FixedSource<C> {
    fn get_mac_key_share(&self) -> Scalar<C> {
        Self::key_share(self.party)
    }
    fn next_local_input_mask(&mut self) -> (Scalar<C>, ScalarShare<C>) {
        let i = self.next(3); // one mask sequence: the sender's local mask i is the receiver's counterparty mask i
        let v = Self::val(3, i);
        (v, self.split(3, i, v))
    }
    fn next_counterparty_input_mask(&mut self) -> ScalarShare<C> {
        let i = self.next(3);
        self.split(3, i, Self::val(3, i))
    }
    fn next_shared_bit(&mut self) -> ScalarShare<C> {
        let i = self.next(4);
        self.split(4, i, Scalar::from(i & 1))
    }
    fn next_shared_value(&mut self) -> ScalarShare<C> {
        let i = self.next(5);
        self.split(5, i, Self::val(5, i))
    }
    fn next_shared_inverse_pair(&mut self) -> (ScalarShare<C>, ScalarShare<C>) {
        let i = self.next(6);
        let v = Self::val(6, i);
        (self.split(6, i, v), self.split(7, i, v.inverse()))
    }
    fn next_triplet(&mut self) -> (ScalarShare<C>, ScalarShare<C>, ScalarShare<C>) {
        let i = self.next(1);
        let (a, b) = (Self::val(1, i), Self::val(2, i));
        (self.split(1, i, a), self.split(2, i, b), self.split(8, i, a * b))
    }
}

/// lib.rs:116-128 execute_mock_mpc, generic over the curve and over FixedSource (the reference's helper is bound to its TestCurve)
async fn run_two_parties<C, T, S, F>(mut f: F) -> (T, T)
where
    C: CurveGroup,
    T: Send + 'static,
    S: Future<Output = T> + Send + 'static,
    F: FnMut(MpcFabric<C>) -> S,
{
    let (s0, s1) = UnboundedDuplexStream::<C>::new_duplex_pair();
    let f0 = MpcFabric::new(MockNetwork::new(PARTY0, s0), FixedSource::<C>::new(PARTY0));
    let f1 = MpcFabric::new(MockNetwork::new(PARTY1, s1), FixedSource::<C>::new(PARTY1));
    let t0 = tokio::spawn(f(f0.clone()));
    let t1 = tokio::spawn(f(f1.clone()));
    let (o0, o1) = (t0.await.unwrap(), t1.await.unwrap());
    f0.shutdown();
    f1.shutdown();
    (o0, o1)
}

fn share_json<C: CurveGroup>(s: &ScalarShare<C>) -> Value {
    json!([dec(&s.share()), dec(&s.mac())])
}
fn point_share_json<C: CurveGroup>(s: &PointShare<C>) -> Value {
    json!([hex(&s.share().to_bytes()), hex(&s.mac().to_bytes())])
}

/// AuthenticatedScalarResult::batch_mul (authenticated_scalar.rs:848-879) under FixedSource on curve C: every party's LOCAL shares of the
/// inputs, of the triples the gate consumed and of the products, and the authenticated opening.  C = Curve25519 is BASELINE config 1's
/// field, C = BLS12-381 config 5's.
async fn batch_mul_fixed_vectors<C: CurveGroup>(curve: &'static str) -> Value
where
    C::ScalarField: Unpin,
{
    let xs: Vec<String> = test_scalars::<C>().iter().map(dec).collect();
    let ys: Vec<String> = test_scalars::<C>().iter().rev().map(dec).collect();
    let (xs2, ys2) = (xs.clone(), ys.clone());
    let (p0, p1) = run_two_parties::<C, _, _, _>(move |fabric| {
        let (xs, ys) = (xs2.clone(), ys2.clone());
        async move {
            let party = fabric.party_id();
            let n = xs.len();
            let x: Vec<Scalar<C>> = xs.iter().map(|d| scalar_from_dec(d)).collect();
            let y: Vec<Scalar<C>> = ys.iter().map(|d| scalar_from_dec(d)).collect();
            let a = fabric.batch_share_scalar(x, PARTY0); // fabric.rs:578-600: masks 0..n
            let b = fabric.batch_share_scalar(y, PARTY1); //                     masks n..2n
            let prod = AuthenticatedScalarResult::batch_mul(&a, &b); // triples 0..n (fabric.rs:894-915)
            // the triples the gate pulled: the same closed formulas, from a fresh source of this party
            let (ta, tb, tc) = FixedSource::<C>::new(party).next_triplet_batch(n);
            let mut rows = Vec::new();
            for i in 0..n {
                let (sa, sb, sp) = (a[i].clone().await, b[i].clone().await, prod[i].clone().await);
                rows.push(json!({
                    "x_share": share_json(&sa), "y_share": share_json(&sb),
                    "a_share": share_json(&ta[i]), "b_share": share_json(&tb[i]), "c_share": share_json(&tc[i]),
                    "product_share": share_json(&sp),
                }));
            }
            let opened = AuthenticatedScalarResult::open_authenticated_batch(&prod);
            let mut vals = Vec::new();
            for o in opened {
                vals.push(dec(&o.await.expect("MAC check")));
            }
            json!({ "key_share_dec": dec(&fabric.mac_key()), "shares": rows, "opened_dec": vals })
        }
    })
    .await;
    json!({ "curve": curve, "source": "FixedSource", "x_dec": xs, "y_dec": ys, "party0": p0, "party1": p1 })
}

/// open_authenticated_batch (authenticated_scalar.rs:278-354) on curve C under FixedSource, with its intermediates: every party's local shares,
/// the opened values the protocol returned, and -- restated with the public Scalar methods the reference's closures call, since the blinder is
/// drawn inside HashCommitmentResult::batch_commit (commitment.rs:67-68) and cannot be injected -- this party's MAC-check shares
/// mac_key * v_i - mac_i (:299-311) and their commitment under a FIXED blinder (commitment.rs:71-86).
async fn open_authenticated_vectors<C: CurveGroup>(curve: &'static str) -> Value
where
    C::ScalarField: Unpin,
{
    let vs: Vec<String> = test_scalars::<C>().iter().map(dec).collect();
    let vs2 = vs.clone();
    let (p0, p1) = run_two_parties::<C, _, _, _>(move |fabric| {
        let vs = vs2.clone();
        async move {
            let party = fabric.party_id();
            let v: Vec<Scalar<C>> = vs.iter().map(|d| scalar_from_dec(d)).collect();
            let shared = fabric.batch_share_scalar(v, PARTY1);
            let mut local = Vec::new();
            for s in shared.iter() {
                local.push(s.clone().await);
            }
            let opened = AuthenticatedScalarResult::open_authenticated_batch(&shared);
            let mut vals = Vec::new();
            for o in opened {
                vals.push(o.await.expect("MAC check"));
            }
            let key = fabric.mac_key();
            let chk: Vec<Scalar<C>> = vals.iter().zip(local.iter()).map(|(v, s)| key * *v - s.mac()).collect();
            let blinder = Scalar::<C>::from(0xB11D0000u64 + party);
            let mut hasher = Sha3_256::new();
            for c in &chk {
                hasher.update(c.to_bytes_be());
            }
            hasher.update(blinder.to_bytes_be());
            let commitment = Scalar::<C>::from_be_bytes_mod_order(&hasher.finalize());
            json!({
                "key_share_dec": dec(&key),
                "shares": local.iter().map(share_json).collect::<Vec<_>>(),
                "opened_dec": vals.iter().map(dec).collect::<Vec<_>>(),
                "mac_check_shares_dec": chk.iter().map(dec).collect::<Vec<_>>(),
                "blinder_dec": dec(&blinder),
                "mac_check_commitment_dec": dec(&commitment),
            })
        }
    })
    .await;
    json!({ "curve": curve, "source": "FixedSource", "values_dec": vs, "party0": p0, "party1": p1 })
}

/// BASELINE config 4 on curve C under FixedSource: points P_i = s_i * G shared as PointShares; `PointShare x public Scalar`
/// (AuthenticatedPointResult::batch_mul_public, authenticated_curve.rs:718-751 -> curve/share.rs:108-114) and the full Beaver
/// AuthenticatedPointResult::batch_mul (:682-714).  Every local PointShare as the compressed bytes of its two points (curve.rs:103-108).
async fn point_mul_vectors<C: CurveGroup>(curve: &'static str) -> Value
where
    C::ScalarField: Unpin,
{
    let ss: Vec<String> = test_scalars::<C>().iter().skip(1).take(8).map(dec).collect(); // P_i = s_i * G
    let ks: Vec<String> = test_scalars::<C>().iter().rev().take(8).map(dec).collect(); // public multipliers
    let xs: Vec<String> = test_scalars::<C>().iter().skip(5).take(8).map(dec).collect(); // shared multipliers
    let (ss2, ks2, xs2) = (ss.clone(), ks.clone(), xs.clone());
    let (p0, p1) = run_two_parties::<C, _, _, _>(move |fabric| {
        let (ss, ks, xs) = (ss2.clone(), ks2.clone(), xs2.clone());
        async move {
            let party = fabric.party_id();
            let n = ss.len();
            let g = CurvePoint::<C>::generator();
            let pts: Vec<CurvePoint<C>> = ss.iter().map(|d| g * scalar_from_dec::<C>(d)).collect();
            let k: Vec<Scalar<C>> = ks.iter().map(|d| scalar_from_dec(d)).collect();
            let x: Vec<Scalar<C>> = xs.iter().map(|d| scalar_from_dec(d)).collect();
            let shared_pts = fabric.batch_share_point(pts, PARTY0); // fabric.rs:622-649: masks 0..n
            let shared_x = fabric.batch_share_scalar(x, PARTY1); //                        masks n..2n
            let k_pub = fabric.allocate_scalars(k);
            let mul_pub = AuthenticatedPointResult::batch_mul_public(&k_pub, &shared_pts);
            let beaver = AuthenticatedPointResult::batch_mul(&shared_x, &shared_pts); // triples 0..n
            let (ta, tb, tc) = FixedSource::<C>::new(party).next_triplet_batch(n);
            let mut rows = Vec::new();
            for i in 0..n {
                let (sp, sx) = (shared_pts[i].clone().await, shared_x[i].clone().await);
                let (mp, bm) = (mul_pub[i].clone().await, beaver[i].clone().await);
                rows.push(json!({
                    "point_share": point_share_json(&sp), "x_share": share_json(&sx),
                    "a_share": share_json(&ta[i]), "b_share": share_json(&tb[i]), "c_share": share_json(&tc[i]),
                    "mul_public_share": point_share_json(&mp), "beaver_mul_share": point_share_json(&bm),
                }));
            }
            let mut opened_pub = Vec::new();
            for o in AuthenticatedPointResult::open_authenticated_batch(&mul_pub) {
                opened_pub.push(hex(&o.await.expect("MAC check").to_bytes()));
            }
            let mut opened_beaver = Vec::new();
            for o in AuthenticatedPointResult::open_authenticated_batch(&beaver) {
                opened_beaver.push(hex(&o.await.expect("MAC check").to_bytes()));
            }
            json!({ "key_share_dec": dec(&fabric.mac_key()), "rows": rows, "opened_mul_public_hex": opened_pub, "opened_beaver_mul_hex": opened_beaver })
        }
    })
    .await;
    json!({ "curve": curve, "source": "FixedSource", "point_scalars_dec": ss, "public_scalars_dec": ks, "shared_scalars_dec": xs, "party0": p0, "party1": p1 })
}

#[tokio::main]
async fn main() {
    let mut scalars = scalar_vectors::<Bn254>("bn254");
    scalars.extend(scalar_vectors::<Ed25519>("curve25519"));
    scalars.extend(scalar_vectors::<Bls12_381>("bls12_381"));
    let mut points = point_vectors::<Bn254>("bn254");
    points.extend(point_vectors::<Ed25519>("curve25519"));
    let mut wire = wire_vectors::<Bn254>("bn254", true);
    wire.extend(wire_vectors::<Ed25519>("curve25519", true));
    wire.extend(wire_vectors::<Bls12_381>("bls12_381", false));
    let mut commitments = commitment_vectors::<Bn254>("bn254");
    commitments.extend(commitment_vectors::<Ed25519>("curve25519"));
    commitments.extend(commitment_vectors::<Bls12_381>("bls12_381"));
    let doc = json!({
        "generator": "tools/ref_vectors: renegade-fi/ark-mpc (online-phase) over ark-* 0.4, sha3 0.10, serde_json 1",
        "schema": 2,
        // the commit of ark-mpc the vectors came from (Dockerfile: the resolved ARK_MPC_REV; a bare cargo run: set it yourself)
        "ark_mpc_rev": std::env::var("ARK_MPC_REV_RESOLVED").unwrap_or_else(|_| "unrecorded".to_string()),
        "scalars": scalars,
        "points": points,
        "wire": wire,
        "commitments": commitments,
        "batch_mul": batch_mul_vectors().await,
        // schema 2: non-degenerate preprocessing, all three scalar fields of the BASELINE configs, config 5's intermediates, config 4's point gates
        "batch_mul_fixed": [
            batch_mul_fixed_vectors::<Bn254>("bn254").await,
            batch_mul_fixed_vectors::<Ed25519>("curve25519").await,
            batch_mul_fixed_vectors::<Bls12_381>("bls12_381").await,
        ],
        "open_authenticated": [
            open_authenticated_vectors::<Bls12_381>("bls12_381").await,
            open_authenticated_vectors::<Bn254>("bn254").await,
        ],
        "point_mul": [
            point_mul_vectors::<Bn254>("bn254").await,
            point_mul_vectors::<Ed25519>("curve25519").await,
        ],
    });
    println!("{}", serde_json::to_string_pretty(&doc).unwrap());
}
